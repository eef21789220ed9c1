"""Deterministic, transcendental-free weight / input generator.

Every value is a pure function of (seed, tensor name, flat index): a splitmix64
finaliser over 64-bit integers, four 24-bit uniforms summed into an
Irwin-Hall(4) variate (mean 0, unit variance after scaling).  Only integer ops,
adds and multiplies are used, so the same fp32 bits come out on every machine:
the golden-vector harness (tests/golden/make_golden.py, which fills the imported
reference model) and the build (tests, bench.py, smoke) share weights without
shipping weight files.

Scale rules by state_dict key follow the reference's init where it matters for
shape only ("weights: deterministic hash init"); stds are
chosen larger than BERT's 0.02 so that attention / pooling softmaxes are far
from uniform and the parity tests exercise them.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def _key(seed, name):
    return np.uint64((int(seed) * 0x100000001B3 + zlib.crc32(name.encode())) & 0xFFFFFFFFFFFFFFFF)


def hash_u64(seed, name, n):
    """n 64-bit hashes for tensor `name`."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix(_splitmix(idx ^ _key(seed, name)) + np.uint64(0x632BE59BD9B4E019))


def hash_normal(seed, name, shape, std=1.0, mean=0.0):
    """Irwin-Hall(4) approximate normal, exact-arithmetic fp32."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        h1 = hash_u64(seed, name, n)
        h2 = _splitmix(h1 ^ np.uint64(0xD6E8FEB86659FD93))
    m24 = np.uint64(0xFFFFFF)
    s = ((h1 & m24).astype(np.float64) + ((h1 >> np.uint64(24)) & m24).astype(np.float64)
         + (h2 & m24).astype(np.float64) + ((h2 >> np.uint64(24)) & m24).astype(np.float64))
    s = s / 16777216.0 - 2.0                       # mean 0, var 4/12
    v = s * (1.7320508075688772 * std) + mean      # sqrt(3) -> unit variance
    return v.astype(np.float32).reshape(shape)


def hash_uniform(seed, name, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    h = hash_u64(seed, name, n)
    u = (h >> np.uint64(40)).astype(np.float64) / 16777216.0
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def hash_randint(seed, name, shape, lo, hi):
    """Integers in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = hash_u64(seed, name, n)
    return (lo + (h >> np.uint64(11)) % np.uint64(hi - lo)).astype(np.int64).reshape(shape)


# (suffix match, kind, std/param) -- first hit wins
_RULES = [
    ("LayerNorm.weight", "normal", (0.1, 1.0)),
    ("LayerNorm.bias", "normal", (0.1, 0.0)),
    ("word_embeddings.weight", "normal", (0.5, 0.0)),
    ("position_embeddings.weight", "normal", (0.2, 0.0)),
    ("token_type_embeddings.weight", "normal", (0.2, 0.0)),
    ("rel_pos_bias.weight", "normal", (0.5, 0.0)),
    ("self.key.bias", "normal", (0.05, 0.0)),
    ("att_fc1.weight", "normal", (0.05, 0.0)),
    ("att_fc1.bias", "normal", (0.05, 0.0)),
    ("att_fc2.weight", "normal", (0.2, 0.0)),
    ("att_fc2.bias", "normal", (0.05, 0.0)),
    ("pad_doc", "uniform", (-1.0, 1.0)),
    ("news_encoder.dense.weight", "normal", (0.005, 0.0)),
    ("multi_head_self_attn.W_Q.weight", "fan", (2.0, 0.0)),        # NRMS: std = 2/sqrt(fan_in): attention far enough from
    ("multi_head_self_attn.W_K.weight", "fan", (2.0, 0.0)),        # uniform to carry gradient, raw exp() still tame
    ("multi_head_self_attn.W_V.weight", "fan", (1.0, 0.0)),
    (".bias", "normal", (0.05, 0.0)),
    (".weight", "normal", (0.05, 0.0)),
]


def init_tensor(seed, name, shape):
    """fp32 ndarray for state_dict entry `name` (keys follow SURVEY.md 8-b)."""
    for suffix, kind, prm in _RULES:
        if name.endswith(suffix):
            if kind == "uniform":
                return hash_uniform(seed, name, shape, prm[0], prm[1])
            if kind == "fan":
                return hash_normal(seed, name, shape, std=prm[0] / float(shape[-1]) ** 0.5, mean=prm[1])
            return hash_normal(seed, name, shape, std=prm[0], mean=prm[1])
    return hash_normal(seed, name, shape, std=0.05)


def init_state_dict(seed, shapes):
    """shapes: {key: tuple} -> {key: fp32 ndarray}."""
    return {k: init_tensor(seed, k, tuple(s)) for k, s in shapes.items()}


def pretrained_like(sd, seed):
    """Give a hash-initialised state dict (in place, returned) the statistics a PRETRAINED UniLM / BERT checkpoint has and a
    std-0.02 random init lacks - the ones that decide whether 16-bit activations and a loss-scaled 16-bit backward hold up on real
    weights (unilm2-base-uncased.bin itself is not available offline): a few LayerNorm channels with gamma 12 ... 30 x the rest
    (and one nearly switched off), shifted beta on the same channels, a handful of embedding rows 6 x larger than the others,
    two FFN output-bias channels far off zero - hidden states reach |h| ~ 100 in the outlier channels, as in published
    checkpoints.  Pure function of (seed, key): the golden harness (tests/golden/make_golden.py) applies it to the reference
    model's weights, tests/helpers.py to the build's."""
    for k in sorted(sd):
        v = sd[k]
        if k.endswith("LayerNorm.weight") or k.endswith("LayerNorm.bias"):
            ch = hash_randint(seed, "outlier." + k.rsplit(".", 1)[0], (4,), 0, v.shape[0])
            if k.endswith("weight"):
                v[ch] = v[ch] * np.array([12.0, 20.0, 30.0, 0.05], np.float32)
            else:
                v[ch] = v[ch] + np.array([1.5, -2.0, 0.0, 0.0], np.float32)
        elif k.endswith("word_embeddings.weight"):
            rows = hash_randint(seed, "outlier." + k, (16,), 1, v.shape[0])
            v[rows] = v[rows] * np.float32(6.0)
        elif k.endswith("output.dense.bias") and ".attention." not in k:
            ch = hash_randint(seed, "outlier." + k, (2,), 0, v.shape[0])
            v[ch] = v[ch] + np.array([3.0, -3.0], np.float32)
    return sd
