"""Synthetic MIND-shaped inputs: there is no network for MIND itself.

news table: n rows + pad row 0; title length ~ clipped N(14, 4) word pieces incl. [CLS]=101 / [SEP]=102,
ids uniform in [1000, 30521], right-padded with 0 to L, mask 1/0 (preprocess.py:32-33,61-62);
histories: length ~ min(U, geometric mean ~32) left-padded with the pad news, ids Zipf(1.1) over news;
candidates: 1 positive + npratio negatives uniform; label uniform in [0, npratio] (dataloader.py:136-137);
teacher tables T x (n+1, D) fp32.  Everything is a function of the seed (hashinit integer hash)."""
import numpy as np

import hashinit


def news_table(seed, n_news, L, vocab=30522, mean_len=14.0, std_len=4.0):
    u = hashinit.hash_normal(seed, "synth.len", (n_news + 1,), std=std_len, mean=mean_len)
    lens = np.clip(np.rint(u), 3, L).astype(np.int64)
    ids = hashinit.hash_randint(seed, "synth.ids", (n_news + 1, L), 1000, vocab)
    pos = np.arange(L)[None, :]
    mask = (pos < lens[:, None]).astype(np.int32)
    ids = ids * mask
    ids[:, 0] = 101
    ids[np.arange(n_news + 1), lens - 1] = 102
    comb = np.concatenate([ids.astype(np.int32), mask], 1)
    comb[0] = 0                                    # row 0 = the all-zero pad news (preprocess.py:48-66)
    return np.ascontiguousarray(comb)


def impressions(seed, n_imp, n_news, U, C):
    """-> hist_idx (n,U) int32, mask (n,U) f32, cand_idx (n,C) int32, label (n,) int64"""
    g = hashinit.hash_uniform(seed, "synth.hl", (n_imp,), 1e-6, 1.0)
    hl = np.minimum(U, np.floor(np.log(g) / np.log(1.0 - 1.0 / 32.0)).astype(np.int64))
    z = hashinit.hash_uniform(seed, "synth.zipf", (n_imp, U), 1e-9, 1.0)
    # inverse-CDF of a truncated Zipf(1.1) by a power-law approximation (exact law is irrelevant for timing)
    hidx = 1 + np.minimum(n_news - 1, np.floor(n_news * z ** 10.0)).astype(np.int64)
    slot = np.arange(U)[None, :]
    valid = slot >= (U - hl)[:, None]
    hidx = np.where(valid, hidx, 0)
    cidx = hashinit.hash_randint(seed, "synth.cand", (n_imp, C), 1, n_news + 1)
    label = hashinit.hash_randint(seed, "synth.label", (n_imp,), 0, C)
    return hidx.astype(np.int32), valid.astype(np.float32), cidx.astype(np.int32), label.astype(np.int64)


def teacher_tables(seed, T, n_news, D, std=0.3):
    return np.stack([hashinit.hash_normal(seed, "synth.teacher%d" % i, (n_news + 1, D), std=std) for i in range(T)], 0)
