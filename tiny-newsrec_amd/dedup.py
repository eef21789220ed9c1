"""In-batch de-duplication plan (host side, integer-exact).

A training batch names B*(U+C) news slots, but many of them are the same news: every left-padded history slot is
news 0 (dataloader.py:73-83) and popular news recur across impressions.  Identical token rows encode to identical
vectors (SURVEY.md appendix iii; in this engine bit-identical, rows are processed independently), so the encoder
only needs to run once per DISTINCT news id of the step; the vectors are expanded back to the slots before the user
encoder / scorer / KD losses, and the slot gradients are summed per distinct id before the encoder backward.
FLOP accounting (bench.py) stays un-deduplicated as SURVEY.md 8-d prescribes.

The plan is built by the loader's producer thread next to the sample decode (numpy, ~50 us for 1760 slots):
    uniq  (n_enc,)   int32  distinct news ids, padded with id 0 to a multiple of `quantum` sequences so that the
                            token count stays a multiple of 64 and only a few distinct workspace shapes occur
    inv   (N,)       int32  slot -> row of uniq
    order (N,)       int32  slots grouped by row of uniq (stable), seg (n_enc+1,) int32 the group offsets
"""
import numpy as np


class DedupPlan:
    __slots__ = ("uniq", "inv", "order", "seg", "n_enc", "n_unique", "n_slots")

    def to(self, device):
        import torch
        out = DedupPlan()
        for k in ("uniq", "inv", "order", "seg"):
            v = getattr(self, k)
            setattr(out, k, torch.from_numpy(v).to(device, non_blocking=True) if isinstance(v, np.ndarray) else v.to(device))
        out.n_enc, out.n_unique, out.n_slots = self.n_enc, self.n_unique, self.n_slots
        return out


def build_plan(hist_idx, cand_idx, quantum=64):
    """hist_idx (B,U), cand_idx (B,C) integer arrays -> DedupPlan (numpy members), or None when nothing would be
    saved (padded distinct count >= slot count)."""
    slots = np.concatenate([np.asarray(hist_idx).reshape(-1), np.asarray(cand_idx).reshape(-1)]).astype(np.int64)
    n = slots.size
    uniq, inv = np.unique(slots, return_inverse=True)
    n_unique = int(uniq.size)
    n_enc = -(-n_unique // quantum) * quantum
    if n_enc >= n:
        return None
    p = DedupPlan()
    p.uniq = np.zeros(n_enc, dtype=np.int32)
    p.uniq[:n_unique] = uniq
    p.inv = inv.reshape(-1).astype(np.int32)
    p.order = np.argsort(p.inv, kind="stable").astype(np.int32)
    seg = np.zeros(n_enc + 1, dtype=np.int64)
    np.cumsum(np.bincount(p.inv, minlength=n_enc), out=seg[1:])
    p.seg = seg.astype(np.int32)
    p.n_enc, p.n_unique, p.n_slots = n_enc, n_unique, n
    return p
