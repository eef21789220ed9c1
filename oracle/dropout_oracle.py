"""ORACLE -- test infrastructure only, never a product path.

Counter-based dropout masks of the HIP path restated in numpy, bit for bit (integer work): Philox4x32-10 (Salmon et al.,
"Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator torch's CUDA dropout draws from as well) keyed by the
run's seed, counter = (call index low, call index high, site, forward-call number); one call yields 8 16-bit uniforms, an
element is KEPT iff its uniform >= floor(p * 65536 + 0.5) and kept elements are scaled by 1 / (1 - p) as
torch.nn.functional.dropout does (tnlrv3/modeling.py:177, 224; transformers BertSelfOutput / BertOutput dropout, call sites
tnlrv3/modeling.py:287, 306).  The reference draws its masks from torch's global generator, which no other implementation
can reproduce; parity is therefore "same mask -> same numbers": tests/golden/make_golden.py runs the reference with its
nn.Dropout modules replaced by these masks, the oracle and the HIP path consume the same ones.

Sites (tiny-newsrec_amd/csrc/dropout.h): kind | layer << 8, kinds 0 embeddings (N*L, H), 1 attention probabilities
(N, A, Lr, Lr), 2 attention-output dense (N*L, H), 3 FFN-output dense (N*L, H).
Row-major (rows, cols) sites: element index e = row * cols + col, call index e >> 3, uniform e & 7 of the call.
Attention probabilities: 4 x 4 blocks of (query, key); call index ((pair * nb + q >> 2) * nb + k >> 2) * 2 + ((q & 3) >> 1)
with pair = n * A + a and nb = Lr / 4, uniform ((q & 1) << 2) | (k & 3) -- a lane of either kernel orientation (one query x
4 keys, or 4 queries x one key) then needs one or two calls per block."""
import numpy as np

KIND_EMB, KIND_PROB, KIND_ATTN_OUT, KIND_FFN_OUT = 0, 1, 2, 3
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_LO = np.uint64(0xFFFFFFFF)


def site_id(kind, layer=0):
    return int(kind) | (int(layer) << 8)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy arrays of counters (uint32 values held in uint64) -> four uint64 arrays of 32-bit outputs."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & _LO for c in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for r in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _LO, p1 >> np.uint64(32), p1 & _LO
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def threshold(p):
    return int(np.floor(float(p) * 65536.0 + 0.5))


def _uniform16(call_idx, sub, seed, site, call):
    """16-bit uniform number `sub` (0..7) of Philox call `call_idx` (uint64 arrays of equal shape)."""
    call_idx = np.asarray(call_idx, dtype=np.uint64)
    o = philox4x32_10(call_idx & _LO, call_idx >> np.uint64(32), np.uint64(site), np.uint64(call), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    sub = np.asarray(sub, dtype=np.uint64)
    word = np.choose((sub >> np.uint64(1)).astype(np.int64), o)
    return (word >> (np.uint64(16) * (sub & np.uint64(1)))) & np.uint64(0xFFFF)


def rows_mask(p, seed, site, call, rows, cols):
    """(rows, cols) fp32 multiplier: 0 for dropped elements, 1 / (1 - p) for kept ones."""
    e = np.arange(rows * cols, dtype=np.uint64)
    u = _uniform16(e >> np.uint64(3), e & np.uint64(7), int(seed), site, call)
    keep = u >= np.uint64(threshold(p))
    return (keep.astype(np.float32) * np.float32(1.0 / (1.0 - p))).reshape(rows, cols)


def probs_mask(p, seed, site, call, n_seq, heads, L):
    """(n_seq, heads, L, L) fp32 multiplier for the attention probabilities (query, key), tile pitch Lr = roundup(L, 32)."""
    Lr = (L + 31) // 32 * 32
    nb = np.uint64(Lr // 4)
    pair = np.arange(n_seq * heads, dtype=np.uint64)[:, None, None]
    q = np.arange(L, dtype=np.uint64)[None, :, None]
    k = np.arange(L, dtype=np.uint64)[None, None, :]
    two, one, three = np.uint64(2), np.uint64(1), np.uint64(3)
    ci = ((pair * nb + (q >> two)) * nb + (k >> two)) * two + ((q & three) >> one)
    sub = ((q & one) << two) | (k & three)
    ci, sub = np.broadcast_arrays(ci, sub)
    u = _uniform16(ci, sub, int(seed), site, call)
    keep = u >= np.uint64(threshold(p))
    return (keep.astype(np.float32) * np.float32(1.0 / (1.0 - p))).reshape(n_seq, heads, L, L)


class Dropout:
    """What an encoder pass needs: p for the hidden-state sites and for the attention probabilities, the seed and this
    forward call's number.  mask_*() return the fp32 multipliers (generated once, cached for the backward)."""

    def __init__(self, p_hidden, p_attn, seed, call):
        self.p_hidden, self.p_attn, self.seed, self.call = float(p_hidden), float(p_attn), int(seed), int(call)

    def hidden(self, kind, layer, rows, cols):
        if self.p_hidden <= 0.0:
            return None
        return rows_mask(self.p_hidden, self.seed, site_id(kind, layer), self.call, rows, cols)

    def probs(self, layer, n_seq, heads, L):
        if self.p_attn <= 0.0:
            return None
        return probs_mask(self.p_attn, self.seed, site_id(KIND_PROB, layer), self.call, n_seq, heads, L)
