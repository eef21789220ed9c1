"""ORACLE -- test infrastructure only, never a product path.

Pure-Python restatement of the integer (bit-exact) side of the hot path: the
data-parallel file-sharding rule and the per-batch sample decoding.  Pinned by
tests/golden/datapath.npz, captured from the imported reference
(tests/golden/make_golden.py).  Citations: /root/reference/Tiny-NewsRec/.
"""
import fnmatch
import os
import random

import numpy as np


def get_worker_files(dirname, worker_rank, world_size, filename_pat="*", shuffle=False, seed=0):
    """streaming.py:40-58: sorted matches, optional seeded shuffle, files[rank::world]."""
    files = sorted(os.path.join(dirname, x) for x in os.listdir(dirname)
                   if not os.path.isdir(os.path.join(dirname, x)) and fnmatch.fnmatch(x, filename_pat))
    if shuffle:
        random.seed(seed)
        random.shuffle(files)
    return files[worker_rank::world_size]


def trans_to_nindex(news_index, nids):
    """dataloader.py:73-74: unknown id -> 0."""
    return [news_index[i] if i in news_index else 0 for i in nids]


def pad_to_fix_len(x, fix_length, padding_front=True, padding_value=0):
    """dataloader.py:76-83: keep the LAST fix_length entries, left-pad; mask 0..0 1..1."""
    if padding_front:
        pad_x = [padding_value] * (fix_length - len(x)) + x[-fix_length:]
        mask = [0] * (fix_length - len(x)) + [1] * min(fix_length, len(x))
    else:
        pad_x = x[-fix_length:] + [padding_value] * (fix_length - len(x))
        mask = [1] * min(fix_length, len(x)) + [0] * (fix_length - len(x))
    return pad_x, mask


def decode_batch(lines, news_index, user_log_length, npratio, labels=None, rng=random):
    """dataloader.py:118-149, index level: TSV lines -> (hist_idx (B,U), mask (B,U), cand_idx (B,C), label (B,)).

    labels=None draws label = rng.randint(0, npratio) per line exactly like :136."""
    H, Mk, Cd, Lb = [], [], [], []
    for n, raw in enumerate(lines):
        line = raw.decode("utf-8").split("\t") if isinstance(raw, bytes) else raw.split("\t")
        click, mask = pad_to_fix_len(trans_to_nindex(news_index, line[3].split()), user_log_length)
        pos = trans_to_nindex(news_index, line[4].split())
        neg = trans_to_nindex(news_index, line[5].split())
        label = rng.randint(0, npratio) if labels is None else int(labels[n])
        H.append(click); Mk.append(mask); Cd.append(neg[:label] + pos + neg[label:]); Lb.append(label)
    return (np.asarray(H, np.int64), np.asarray(Mk, np.float32), np.asarray(Cd, np.int64),
            np.asarray(Lb, np.int64))


def gather_batch(hist_idx, mask, cand_idx, label, news_combined, teacher_embs):
    """dataloader.py:131,138,140-144,162-170: the 6-tuple run.py:175 consumes."""
    return (news_combined[hist_idx].astype(np.int64), mask.astype(np.float32),
            news_combined[cand_idx].astype(np.int64), label.astype(np.int64),
            [t[hist_idx].astype(np.float32) for t in teacher_embs],
            [t[cand_idx].astype(np.float32) for t in teacher_embs])


def build_news_combined(titles, num_words_title):
    """preprocess.py:48-66 + run.py:53: row 0 = all-zero pad news; row i = ids ++ attention mask, int32.

    titles: list (news order, index 1..n) of (input_ids, attention_mask) already padded/truncated."""
    n = len(titles) + 1
    out = np.zeros((n, 2 * num_words_title), dtype="int32")
    for i, (ids, am) in enumerate(titles, start=1):
        out[i, :num_words_title] = ids
        out[i, num_words_title:] = am
    return out
