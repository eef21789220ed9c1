"""GPU parity, kernel by kernel: every C-ABI entry point against the numpy oracle on seeded inputs.

Integer-valued operands give bit-exact checks of the MFMA fragment / LDS-swizzle / transposed-read
layouts (asymmetric data, so a row<->col swap cannot hide); random operands check the fused epilogues.
Tolerances: bf16 outputs carry one rounding of 2^-9 relative -> rtol 1e-2 on bf16 tensors; the fp16 build (the headline
dtype) is held to 1.5e-3 (LayerNorm) and 2e-3 / 3e-3 (attention forward / backward) on the same cases; fp32 heads 1e-4."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tnr_hip as T                      # noqa: E402
from oracle import newsrec_oracle as O   # noqa: E402

DEV = "cuda:0"


def dev(x, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t.to(dt) if dt is not None else t


def bf(x):
    """numpy fp32 -> bf16-rounded fp32 (round to nearest even), what the kernels see."""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def hf(x):
    """numpy fp32 -> fp16-rounded fp32: what the _f16 build's kernels see."""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.float16).float().numpy()


# the two 16-bit builds: (rounding of test inputs, torch dtype, entry-point suffix)
BUILDS = {"bf16": (bf, torch.bfloat16, ""), "fp16": (hf, torch.float16, "_f16")}


def rnd(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def test_library_loads():
    assert T.query("tnr_version") == 1


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (130, 256, 128), (257, 128, 192), (1000, 768, 768)])
def test_gemm_nt_integer_exact(M, N, K):
    rs = np.random.RandomState(M + N + K)
    A = rs.randint(-3, 4, (M, K)).astype(np.float32)
    B = rs.randint(-3, 4, (N, K)).astype(np.float32)
    B[:, 0] += np.arange(N) % 5          # asymmetric
    A[:, 1] += np.arange(M) % 3
    a, b = dev(A, torch.bfloat16), dev(B, torch.bfloat16)
    c = torch.full((M, N), 7.0, device=DEV, dtype=torch.float32)
    T.call("tnr_gemm_nt", a, K, b, K, c, N, M, N, K, None, None, 0, None, 0, T.EPI_OUTF32)
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy(), A @ B.T)


def test_gemm_nt_epilogues():
    M, N, K = 333, 384, 256
    A, B = bf(rnd((M, K), 1)), bf(rnd((N, K), 2, 0.1))
    bias, res, aux = rnd((N,), 3), bf(rnd((M, N), 4)), bf(rnd((M, N), 5))
    a, b = dev(A, torch.bfloat16), dev(B, torch.bfloat16)
    ref = A @ B.T
    cases = {
        T.EPI_BIAS: ref + bias,
        T.EPI_BIAS | T.EPI_GELU: O.gelu(ref + bias),
        T.EPI_BIAS | T.EPI_TANH | T.EPI_OUTF32: np.tanh(ref + bias),
        T.EPI_BIAS | T.EPI_RES: ref + bias + res,
        T.EPI_RES: ref + res,
        T.EPI_MULDGELU: ref * O.gelu_grad(aux),
        0: ref,
    }
    for flags, want in cases.items():
        out_f32 = bool(flags & T.EPI_OUTF32)
        c = torch.zeros((M, N), device=DEV, dtype=torch.float32 if out_f32 else torch.bfloat16)
        T.call("tnr_gemm_nt", a, K, b, K, c, N, M, N, K, dev(bias), dev(res, torch.bfloat16), N,
               dev(aux, torch.bfloat16), N, flags)
        torch.cuda.synchronize()
        np.testing.assert_allclose(c.float().cpu().numpy(), want, rtol=1e-2 if not out_f32 else 1e-4,
                                   atol=2e-2 if not out_f32 else 1e-4, err_msg="flags=%d" % flags)
    # AUXOUT: pre-activation stored beside the GELU output
    c = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
    u = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
    T.call("tnr_gemm_nt", a, K, b, K, c, N, M, N, K, dev(bias), None, 0, u, N, T.EPI_BIAS | T.EPI_GELU | T.EPI_AUXOUT)
    torch.cuda.synchronize()
    np.testing.assert_allclose(u.float().cpu().numpy(), ref + bias, rtol=1e-2, atol=2e-2)
    np.testing.assert_allclose(c.float().cpu().numpy(), O.gelu(ref + bias), rtol=1e-2, atol=2e-2)


def test_gemm_nt_does_not_touch_rows_past_M():
    M, N, K = 100, 128, 64
    a = dev(rnd((M, K), 1), torch.bfloat16)
    b = dev(rnd((N, K), 2), torch.bfloat16)
    c = torch.full((128, N), 5.0, device=DEV, dtype=torch.bfloat16)
    T.call("tnr_gemm_nt", a, K, b, K, c, N, M, N, K, None, None, 0, None, 0, 0)
    torch.cuda.synchronize()
    assert (c[M:].float() == 5.0).all()


def test_gemm_nt_rejects_bad_shapes():
    a = torch.zeros((8, 64), device=DEV, dtype=torch.bfloat16)
    with pytest.raises(T.TnrError):
        T.call("tnr_gemm_nt", a, 64, a, 64, a, 100, 8, 100, 64, None, None, 0, None, 0, 0)


@pytest.mark.parametrize("M,N,K,splits", [(64, 128, 128, 1), (200, 256, 128, 2), (1000, 128, 384, 4), (3300, 768, 256, 16)])
def test_gemm_tn_wgrad_integer_exact(M, N, K, splits):
    rs = np.random.RandomState(M + N)
    Mp = (M + 63) // 64 * 64
    dY = np.zeros((Mp, N), np.float32)
    X = np.zeros((Mp, K), np.float32)
    dY[:M] = rs.randint(-2, 3, (M, N))
    X[:M] = rs.randint(-2, 3, (M, K))
    X[:M, 0] += np.arange(M) % 3
    dY[:M, 1] += np.arange(M) % 2
    ws = torch.zeros(T.query("tnr_gemm_tn_ws_elems", N, K, splits), device=DEV)
    dW = torch.full((N, K), 3.0, device=DEV)
    T.call("tnr_gemm_tn_wgrad", dev(dY, torch.bfloat16), N, dev(X, torch.bfloat16), K, dW, K, M, N, K, ws, splits, 0)
    torch.cuda.synchronize()
    want = dY.T @ X
    assert np.array_equal(dW.cpu().numpy(), want)
    T.call("tnr_gemm_tn_wgrad", dev(dY, torch.bfloat16), N, dev(X, torch.bfloat16), K, dW, K, M, N, K, ws, splits, 1)
    torch.cuda.synchronize()
    assert np.array_equal(dW.cpu().numpy(), 2 * want)


# ------------------------------------------------------------------------------------------------ rows
def test_relpos_table_exact():
    w = rnd((12, 32), 3)
    for L in (24, 30, 32):
        t = torch.zeros((12, 32, 32), device=DEV)
        T.call("tnr_relpos_table", dev(w), 12, L, t)
        torch.cuda.synchronize()
        got = t.cpu().numpy()
        assert np.array_equal(got[:, :L, :L], O.relpos_bias_table(w, L))
        assert (got[:, L:, :] == 0).all() and (got[:, :, L:] == 0).all()


def test_embed_ln_and_mask():
    N, L, H, V = 37, 30, 768, 500
    rs = np.random.RandomState(0)
    ids = rs.randint(0, V, (N, L))
    mask = (rs.rand(N, L) > 0.3).astype(np.int64)
    mask[3] = 0
    tok = np.concatenate([ids, mask], 1).astype(np.int64)
    P = {O.BERT + "embeddings.word_embeddings.weight": rnd((V, H), 1),
         O.BERT + "embeddings.position_embeddings.weight": rnd((512, H), 2),
         O.BERT + "embeddings.token_type_embeddings.weight": rnd((2, H), 3),
         O.BERT + "embeddings.LayerNorm.weight": 1 + rnd((H,), 4, 0.1),
         O.BERT + "embeddings.LayerNorm.bias": rnd((H,), 5, 0.1)}
    out = torch.zeros((N * L, H), device=DEV, dtype=torch.bfloat16)
    madd = torch.zeros((N, 32), device=DEV)
    T.call("tnr_embed_ln_fwd", dev(tok), N, L, H, *[dev(P[k]) for k in P][:2],
           dev(P[O.BERT + "embeddings.token_type_embeddings.weight"][0]),
           dev(P[O.BERT + "embeddings.LayerNorm.weight"]), dev(P[O.BERT + "embeddings.LayerNorm.bias"]), 1e-12, out, madd)
    torch.cuda.synchronize()
    want = O.embeddings_fwd(P, ids).reshape(N * L, H)
    np.testing.assert_allclose(out.float().cpu().numpy(), want, rtol=1e-2, atol=1e-2)
    m = madd.cpu().numpy()
    assert np.array_equal(m[:, :L], (1.0 - mask.astype(np.float32)) * -10000.0)
    assert (m[:, L:] <= -1e29).all()


@pytest.mark.parametrize("dtype,tol", [("bf16", 1e-2), ("fp16", 1.5e-3)])     # one 16-bit rounding of outputs of magnitude <= ~4
@pytest.mark.parametrize("M", [5, 257])
def test_layernorm_fwd_bwd(M, dtype, tol):
    rd, td, sfx = BUILDS[dtype]
    H = 768
    x, dy = rd(rnd((M, H), 1, 2.0)), rd(rnd((M, H), 2))
    g, b = 1 + rnd((H,), 3, 0.1), rnd((H,), 4, 0.1)
    y = torch.zeros((M, H), device=DEV, dtype=td)
    st = torch.zeros((M, 2), device=DEV)
    T.call("tnr_ln_fwd" + sfx, dev(x, td), dev(g), dev(b), 1e-12, y, st, M, H)
    yref, cache = O.layer_norm_fwd(x, g, b, 1e-12)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.float().cpu().numpy(), yref, rtol=tol, atol=tol)
    dx = torch.zeros((M, H), device=DEV, dtype=td)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    part = torch.zeros(T.query("tnr_ln_bwd_part_elems", M, H), device=DEV)
    dxs = torch.zeros(H, device=DEV)
    T.call("tnr_ln_bwd" + sfx, dev(dy, td), dev(x, td), st, dev(g), dx, dg, db, dxs, part, M, H)
    torch.cuda.synchronize()
    dxr, dgr, dbr = O.layer_norm_bwd(dy, cache, g)
    np.testing.assert_allclose(dx.float().cpu().numpy(), dxr, rtol=tol, atol=tol)
    np.testing.assert_allclose(dg.cpu().numpy(), dgr, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), dbr, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(dxs.cpu().numpy(), dx.float().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3)   # fused bias grad


def test_colsum_and_reduce():
    M, N = 1234, 768
    x = bf(rnd((M, N), 1))
    out = torch.ones(N, device=DEV)
    part = torch.zeros(T.query("tnr_colsum_part_elems", M, N), device=DEV)
    T.call("tnr_colsum", dev(x, torch.bfloat16), N, T.BF16, M, N, out, part, 1)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), 1 + x.sum(0), rtol=1e-4, atol=1e-3)
    xf = rnd((77, 256), 2)
    out2 = torch.zeros(256, device=DEV)
    part2 = torch.zeros(T.query("tnr_colsum_part_elems", 77, 256), device=DEV)
    T.call("tnr_colsum", dev(xf), 256, T.F32, 77, 256, out2, part2, 0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out2.cpu().numpy(), xf.sum(0), rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, mask, rel, N, L, A):
    d = 64
    q, k, v = [qkv[:, i * A * d:(i + 1) * A * d].reshape(N, L, A, d).transpose(0, 2, 1, 3) for i in range(3)]
    s = q @ k.transpose(0, 1, 3, 2) / 8.0 + ((1.0 - mask) * -10000.0)[:, None, None, :] + rel[None]
    p = O.softmax(s.astype(np.float32), -1)
    return q, k, v, p, (p @ v).transpose(0, 2, 1, 3).reshape(N * L, A * d)


# the probabilities and the outputs are rounded to 16 bits once each: 2^-9 (bf16) / 2^-12 (fp16) relative per rounding
@pytest.mark.parametrize("dtype,tol_f,tol_b", [("bf16", 2e-2, 3e-2), ("fp16", 2e-3, 3e-3)])
@pytest.mark.parametrize("N,L,A", [(3, 30, 12), (2, 32, 2), (5, 7, 3), (1, 1, 1)])
def test_attention_fwd_bwd(N, L, A, dtype, tol_f, tol_b):
    rd, td, sfx = BUILDS[dtype]
    d = 64
    rs = np.random.RandomState(N * 100 + L)
    qkv = rd(rnd((N * L, 3 * A * d), 1, 1.0))
    mask = (rs.rand(N, L) > 0.3).astype(np.float32)
    mask[0, :] = 1
    if N > 1:
        mask[1, :] = 0          # all-pad title: finite uniform softmax over -10000
    w = rnd((A, 32), 2, 0.5)
    rel = O.relpos_bias_table(w, L)
    tok = np.concatenate([np.ones((N, L)), mask], 1).astype(np.int64)
    # mask_add / rel table through their own kernels (the real call chain)
    H = 768
    madd = torch.zeros((N, 32), device=DEV)
    scratch = torch.zeros((N * L, H), device=DEV, dtype=td)
    z = torch.zeros((600, H), device=DEV)
    T.call("tnr_embed_ln_fwd" + sfx, dev(tok), N, L, H, z, z, z[0], z[0], z[0], 1e-12, scratch, madd)
    relt = torch.zeros((A, 32, 32), device=DEV)
    T.call("tnr_relpos_table", dev(w), A, L, relt)
    ctx = torch.zeros((N * L, A * d), device=DEV, dtype=td)
    T.call("tnr_attn_l32_fwd" + sfx, dev(qkv, td), madd, relt, ctx, N, L, A)
    torch.cuda.synchronize()
    q, k, v, p, want = _attn_ref(qkv, mask, rel, N, L, A)
    np.testing.assert_allclose(ctx.float().cpu().numpy(), want, rtol=tol_f, atol=tol_f)
    # backward
    dctx = rd(rnd((N * L, A * d), 3))
    dqkv = torch.zeros((N * L, 3 * A * d), device=DEV, dtype=td)
    bpart = torch.zeros((N, 3 * A * d), device=DEV)
    T.call("tnr_attn_l32_bwd" + sfx, dev(qkv, td), madd, relt, dev(dctx, td), dqkv, bpart, N, L, A)
    torch.cuda.synchronize()
    dch = dctx.reshape(N, L, A, d).transpose(0, 2, 1, 3)
    dp = dch @ v.transpose(0, 1, 3, 2)
    dv = p.transpose(0, 1, 3, 2) @ dch
    ds = p * (dp - (dp * p).sum(-1, keepdims=True))
    dq = ds @ k / 8.0
    dk = ds.transpose(0, 1, 3, 2) @ q / 8.0
    back = lambda t: t.transpose(0, 2, 1, 3).reshape(N * L, A * d)
    want_d = np.concatenate([back(dq), back(dk), back(dv)], 1)
    got = dqkv.float().cpu().numpy()
    scale = np.abs(want_d).max()
    np.testing.assert_allclose(got, want_d, rtol=tol_b, atol=tol_b * scale)
    np.testing.assert_allclose(bpart.sum(0).cpu().numpy(), got.sum(0), rtol=1e-3, atol=1e-3 * scale * N * L)   # fused bias grad


# ------------------------------------------------------------------------------------------------ heads
@pytest.mark.parametrize("L", [30, 128])
def test_attpool_fwd_bwd(L):
    N, H, Q, QP = 9, 768, 200, 256
    Lr = (L + 31) // 32 * 32
    y = bf(rnd((N, L, H), 1))
    w1, b1 = rnd((Q, H), 2, 0.05), rnd((Q,), 3, 0.05)
    w2, b2 = rnd((1, Q), 4, 0.2), rnd((1,), 5, 0.05)
    out, c = O.att_pool_fwd(y, w1, b1, w2, b2)
    e = np.zeros((N * L, QP), np.float32)
    e[:, :Q] = c["e"].reshape(N * L, Q)
    nv = torch.zeros((N, H), device=DEV)
    alpha = torch.zeros((N, Lr), device=DEV)
    den = torch.zeros(N, device=DEV)
    yd, ed = dev(y.reshape(N * L, H), torch.bfloat16), dev(e)
    T.call("tnr_attpool_fwd", yd, ed, QP, dev(w2[0]), dev(b2), Q, nv, alpha, den, N, L, H)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nv.cpu().numpy(), out, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(alpha.cpu().numpy()[:, :L], c["w"], rtol=1e-4, atol=1e-6)
    dnv = rnd((N, H), 6)
    dx, g1, gb1, g2, gb2 = O.att_pool_bwd(dnv, c, w1, w2)
    dyd = torch.zeros((N * L, H), device=DEV, dtype=torch.bfloat16)
    dpre = torch.ones((N * L, QP), device=DEV, dtype=torch.bfloat16)
    dw2p, db2p = torch.zeros((N, Q), device=DEV), torch.zeros(N, device=DEV)
    db1p = torch.zeros((N, QP), device=DEV)
    T.call("tnr_attpool_bwd", yd, ed, QP, dev(w2[0]), Q, dev(dnv), alpha, den, dyd, dpre, QP, dw2p, db2p, db1p, N, L, H)
    torch.cuda.synchronize()
    direct = c["w"][..., None] * dnv[:, None, :]
    np.testing.assert_allclose(dyd.float().cpu().numpy(), direct.reshape(N * L, H), rtol=1e-2, atol=1e-3)
    da = c["w"] * ((y * dnv[:, None, :]).sum(-1) - ((y * dnv[:, None, :]).sum(-1) * c["w"]).sum(1, keepdims=True))
    dpre_ref = da[..., None] * w2[0][None, None, :] * (1 - c["e"] ** 2)
    got = dpre.float().cpu().numpy()
    np.testing.assert_allclose(got[:, :Q], dpre_ref.reshape(N * L, Q), rtol=1e-2, atol=1e-2 * np.abs(dpre_ref).max())
    assert (got[:, Q:] == 0).all()
    np.testing.assert_allclose(db1p.sum(0).cpu().numpy(), got.sum(0), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(dw2p.sum(0).cpu().numpy(), g2[0], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(db2p.sum().cpu().numpy(), gb2[0], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("L", [65, 128, 500])
@pytest.mark.parametrize("sfx", ["", "_f16"])
def test_attpool_long_equals_the_one_workgroup_kernels(L, sfx):
    """tnr_attpool_fwd_long / _bwd_long (stage-1 bodies, Post-train_KD.ipynb cell 12: few, long sequences; token-parallel parts per
    64-token chunk / one wave per token) against tnr_attpool_fwd / _bwd on the same operands: every output equal up to fp32
    rounding of the chunked sums (16-bit outputs: equal bits but for values that round differently), L = 65 (a one-token second
    chunk), 128 and 500 (a ragged last chunk), and deterministic run to run."""
    N, H, Q, QP = 5, 768, 200, 256
    td = torch.bfloat16 if sfx == "" else torch.float16
    Lr = (L + 31) // 32 * 32
    y = dev(rnd((N * L, H), 1), td)
    e = torch.zeros((N * L, QP), device=DEV)
    e[:, :Q] = torch.tanh(dev(rnd((N * L, Q), 2, 0.7)))
    w2, b2, dnv = dev(rnd((Q,), 4, 0.2)), dev(rnd((1,), 5, 0.05)), dev(rnd((N, H), 6))
    ws = torch.zeros(T.query("tnr_attpool_long_ws_elems" + sfx, N, L, H, Q, QP), device=DEV)

    def run(long):
        nv, alpha, den = torch.zeros((N, H), device=DEV), torch.full((N, Lr), 7.0, device=DEV), torch.zeros(N, device=DEV)
        dy = torch.zeros((N * L, H), device=DEV, dtype=td)
        dpre = torch.ones((N * L, QP), device=DEV, dtype=td)
        dw2p, db2p, db1p = torch.zeros((N, Q), device=DEV), torch.zeros(N, device=DEV), torch.zeros((N, QP), device=DEV)
        if long:
            T.call("tnr_attpool_fwd_long" + sfx, y, e, QP, w2, b2, Q, nv, alpha, den, ws, N, L, H)
            T.call("tnr_attpool_bwd_long" + sfx, y, e, QP, w2, Q, dnv, alpha, dy, dpre, QP, dw2p, db2p, db1p, ws, N, L, H)
        else:
            T.call("tnr_attpool_fwd" + sfx, y, e, QP, w2, b2, Q, nv, alpha, den, N, L, H)
            T.call("tnr_attpool_bwd" + sfx, y, e, QP, w2, Q, dnv, alpha, den, dy, dpre, QP, dw2p, db2p, db1p, N, L, H)
        torch.cuda.synchronize()
        return [nv, alpha, den, dy.float(), dpre.float(), dw2p, db2p, db1p]
    ref, got, again = run(False), run(True), run(True)
    for a_, b_ in zip(got, again):
        assert torch.equal(a_, b_)
    names = ["nv", "alpha", "den", "dy", "dpre", "dw2_part", "db2_part", "db1_part"]
    for n_, r_, g_ in zip(names, ref, got):
        # db2_part = sum_i alpha_i (dw_i - S) is a cancelling sum (0 in exact arithmetic): its rounding noise is held against the
        # size of its neighbour dw2_part, not against itself
        scale = (float(ref[5].abs().max()) if n_ == "db2_part" else float(r_.abs().max())) + 1e-30
        tol = 1e-5 if n_ in ("nv", "alpha", "den") else (8e-3 if n_ in ("dy", "dpre") else 2e-4)
        assert float((r_ - g_).abs().max()) <= tol * scale, (n_, float((r_ - g_).abs().max()), scale)
    assert (got[1][:, L:] == 0).all() and (got[4][:, Q:] == 0).all()


def test_sgemm_variants():
    M, N, K, Z = 150, 200, 77, 3
    A, B, bias, C0 = rnd((Z, M, K), 1), rnd((Z, N, K), 2), rnd((Z, N), 3), rnd((Z, M, N), 4)
    c = dev(C0.copy())
    T.call("tnr_sgemm", dev(A), K, 1, M * K, None, dev(B), K, 1, N * K, c, N, M * N, dev(bias), N, M, N, K, Z, 0.5, 2.0, 1, None)
    torch.cuda.synchronize()
    want = 0.5 * np.einsum("zmk,znk->zmn", A, B) + bias[:, None, :] + 2.0 * C0
    np.testing.assert_allclose(c.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    # transposed A (dW = dY^T X) : A(m,k) = dY[k, m]
    dY, X = rnd((300, 64), 5), rnd((300, 48), 6)
    c2 = torch.zeros((64, 48), device=DEV)
    T.call("tnr_sgemm", dev(dY), 1, 64, 0, None, dev(X), 1, 48, 0, c2, 48, 0, None, 0, 64, 48, 300, 1, 1.0, 0.0, 1, None)
    torch.cuda.synchronize()
    np.testing.assert_allclose(c2.cpu().numpy(), dY.T @ X, rtol=1e-4, atol=1e-4)
    # split-K, batched (ragged last chunk)
    dYb, Xb = rnd((3, 1000, 64), 7), rnd((3, 1000, 48), 8)
    c3 = torch.zeros((3, 64, 48), device=DEV)
    part = torch.zeros(8 * 3 * 64 * 48, device=DEV)
    T.call("tnr_sgemm", dev(dYb), 1, 64, 1000 * 64, None, dev(Xb), 1, 48, 1000 * 48, c3, 48, 64 * 48, None, 0, 64, 48, 1000, 3,
           1.0, 0.0, 8, part)
    torch.cuda.synchronize()
    np.testing.assert_allclose(c3.cpu().numpy(), np.einsum("zkm,zkn->zmn", dYb, Xb), rtol=1e-4, atol=1e-3)


def _user_params(nm, D, Q, seed):
    return dict(pad=rnd((nm, D), seed), w1=rnd((nm, Q, D), seed + 1, 0.05), b1=rnd((nm, Q), seed + 2, 0.05),
                w2=rnd((nm, Q), seed + 3, 0.2), b2=rnd((nm,), seed + 4, 0.05))


@pytest.mark.parametrize("ulm", [0, 1])
def test_user_score_fwd_bwd(ulm):
    nm, B, U, C, D, Q, R = 3, 5, 50, 5, 256, 200, 400
    rs = np.random.RandomState(ulm)
    vec = rnd((nm, R, D), 1, 0.3)
    hidx = rs.permutation(R)[:B * U].reshape(B, U).astype(np.int32)
    cidx = rs.randint(0, R, (B, C)).astype(np.int32)
    mask = (rs.rand(B, U) > 0.4).astype(np.float32)
    mask[0] = 1
    mask[1] = 0
    pr = _user_params(nm, D, Q, 10)
    user, score = torch.zeros((nm, B, D), device=DEV), torch.zeros((nm, B, C), device=DEV)
    e, alpha, den = torch.zeros((nm, B, U, Q), device=DEV), torch.zeros((nm, B, U), device=DEV), torch.zeros((nm, B), device=DEV)
    dv = {k: dev(v) for k, v in pr.items()}
    epre = np.einsum("zrd,zqd->zrq", np.stack([vec[z][hidx.reshape(-1)] for z in range(nm)], 0), pr["w1"]) + pr["b1"][:, None, :]
    T.call("tnr_user_score_fwd", dev(vec), R, dev(hidx), dev(cidx), dev(mask), dv["pad"], dv["w1"], dv["b1"], dv["w2"],
           dv["b2"], ulm, dev(epre.astype(np.float32)), None if ulm else dev((pr["pad"][:, None, :] * pr["w1"]).sum(-1) + pr["b1"]),
           user, B * D, score, e, alpha, den, nm, B, U, C, D, Q)
    # the same pass with fc1 inside the kernel (epre = NULL): what the engine issues
    user_f, score_f = torch.zeros_like(user), torch.zeros_like(score)
    e_f, alpha_f, den_f = torch.zeros_like(e), torch.zeros_like(alpha), torch.zeros_like(den)
    T.call("tnr_user_score_fwd", dev(vec), R, dev(hidx), dev(cidx), dev(mask), dv["pad"], dv["w1"], dv["b1"], dv["w2"],
           dv["b2"], ulm, None, None, user_f, B * D, score_f, e_f, alpha_f, den_f, nm, B, U, C, D, Q)
    torch.cuda.synchronize()
    for got, ref in ((e_f, e), (alpha_f, alpha), (den_f, den)):
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    caches = []
    for z in range(nm):
        P = {"p.pad_doc": pr["pad"][z][None], "p.attn.att_fc1.weight": pr["w1"][z], "p.attn.att_fc1.bias": pr["b1"][z],
             "p.attn.att_fc2.weight": pr["w2"][z][None], "p.attn.att_fc2.bias": pr["b2"][z:z + 1]}
        uref, c = O.user_encoder_fwd(P, "p.", vec[z][hidx], mask, bool(ulm))
        caches.append((P, c))
        sref = np.einsum("bcd,bd->bc", vec[z][cidx], uref)
        for u_, s_ in ((user, score), (user_f, score_f)):
            np.testing.assert_allclose(u_[z].cpu().numpy(), uref, rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(s_[z].cpu().numpy(), sref, rtol=1e-4, atol=1e-4)
    # backward of model 0
    duser = rnd((B, D), 7)
    P, c = caches[0]
    dnews, G = O.user_encoder_bwd(P, "p.", duser, c)
    dvec = torch.zeros((R, D), device=DEV)
    ps = T.query("tnr_user_bwd_part_stride", D, Q)
    assert ps == 2 * Q + D + 1
    part = torch.zeros((B, ps), device=DEV)
    hv, dhv, dpre = torch.zeros((B * U, D), device=DEV), torch.zeros((B * U, D), device=DEV), torch.zeros((B * U, Q), device=DEV)
    dw1 = torch.zeros((Q, D), device=DEV)
    sg_part = torch.zeros(8 * Q * D, device=DEV)
    w1_0 = dv["w1"][0].contiguous()
    T.call("tnr_user_bwd_pre", dev(vec[0]), dev(hidx), dev(mask), dv["pad"], dv["w2"], ulm, dev(duser), e, alpha, hv, dpre, part,
           B, U, D, Q)
    T.call("tnr_sgemm", dpre, 1, Q, 0, None, hv, 1, D, 0, dw1, D, 0, None, 0, Q, D, B * U, 1, 1.0, 0.0, 8, sg_part)
    T.call("tnr_sgemm", dpre, Q, 1, 0, None, w1_0, 1, D, 0, dhv, D, 0, None, 0, B * U, D, Q, 1, 1.0, 0.0, 1, None)
    T.call("tnr_user_bwd_post", dhv, alpha, dev(duser), dev(mask), dev(hidx), ulm, dvec, part, B, U, D, Q)
    torch.cuda.synchronize()
    hv_ref = vec[0][hidx] if ulm else vec[0][hidx] * mask[..., None] + pr["pad"][0][None, None] * (1.0 - mask[..., None])
    np.testing.assert_allclose(hv.cpu().numpy().reshape(B, U, D), hv_ref, rtol=1e-6, atol=1e-7)
    want = np.zeros((R, D), np.float32)
    np.add.at(want, hidx.reshape(-1), dnews.reshape(-1, D))
    np.testing.assert_allclose(dvec.cpu().numpy(), want, rtol=1e-3, atol=1e-5)
    p = np.concatenate([dw1.cpu().numpy().reshape(-1), part.sum(0).cpu().numpy()])
    o = 0
    for key, n in (("p.attn.att_fc1.weight", Q * D), ("p.attn.att_fc1.bias", Q), ("p.attn.att_fc2.weight", Q),
                   ("p.pad_doc", D), ("p.attn.att_fc2.bias", 1)):
        ref = G[key].reshape(-1)
        if key.endswith("att_fc2.bias"):      # mathematical no-op (cancels in the normaliser): rounding noise only
            assert abs(p[o]) < 1e-3 and abs(ref[0]) < 1e-3
        else:
            np.testing.assert_allclose(p[o:o + n], ref, rtol=2e-3, atol=2e-3 * (np.abs(ref).max() + 1e-6), err_msg=key)
        o += n


@pytest.mark.parametrize("T_,tau", [(4, 1.0), (1, 2.0), (0, 1.0)])
def test_kd_score_loss(T_, tau):
    B, C, coef = 37, 5, 0.2
    rs = np.random.RandomState(T_)
    s, ts = rnd((B, C), 1), rnd((max(T_, 1), B, C), 2)
    y = rs.randint(0, C, B)
    tw = torch.zeros((B, max(T_, 1)), device=DEV)
    dscore, losses = torch.zeros((B, C), device=DEV), torch.zeros(4, device=DEV)
    T.call("tnr_kd_score_loss", dev(s), dev(ts) if T_ else None, dev(y), tau, coef, tw if T_ else None, dscore, losses,
           B, C, T_)
    torch.cuda.synchronize()
    target = O.cross_entropy_rows(s, y).mean()
    oh = np.eye(C, dtype=np.float32)[y]
    g = coef * (O.softmax(s) - oh)
    if T_:
        tl = np.stack([O.cross_entropy_rows(ts[i], y) for i in range(T_)], -1)
        w = O.softmax(-tl, -1)
        mix = np.einsum("tbc,bt->bc", ts[:T_], w)
        pT = O.softmax(mix / tau)
        distill = (-(pT * O.log_softmax(s / tau)).sum(-1)).mean()
        g = g + (O.softmax(s / tau) - pT) / tau
        np.testing.assert_allclose(tw.cpu().numpy(), w, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(losses[0].item(), distill, rtol=1e-4)
    np.testing.assert_allclose(losses[1].item(), target, rtol=1e-4)
    np.testing.assert_allclose(dscore.cpu().numpy(), g / B, rtol=1e-3, atol=1e-6)


def test_kd_embed_loss():
    B, U, C, D, T_ = 4, 50, 5, 256, 3
    nn, rt = B * (U + C), B * (U + C + 1)
    S, P, tw = rnd((rt, D), 1, 0.3), rnd((T_, rt, D), 2, 0.3), O.softmax(rnd((B, T_), 3))
    dS, dP = torch.zeros((rt, D), device=DEV), torch.zeros((T_, rt, D), device=DEV)
    loss, part = torch.zeros(1, device=DEV), torch.zeros(rt, device=DEV)
    T.call("tnr_kd_embed_loss", dev(S), dev(P), dev(tw), loss, dS, dP, part, B, U, C, D, T_)
    torch.cuda.synchronize()
    # back to the reference's (B, U+C, D) news layout + (B, D) user layout
    news = lambda X: np.concatenate([X[:B * U].reshape(B, U, D), X[B * U:nn].reshape(B, C, D)], 1)
    ne = np.stack([((news(S) - news(P[i])) ** 2).mean(-1).mean(-1) for i in range(T_)], -1)
    ue = np.stack([((S[nn:] - P[i][nn:]) ** 2).mean(-1) for i in range(T_)], -1)
    want = (ne * tw).sum(-1).mean() + (ue * tw).sum(-1).mean()
    np.testing.assert_allclose(loss.item(), want, rtol=1e-4)
    bidx = np.concatenate([np.repeat(np.arange(B), U), np.repeat(np.arange(B), C), np.arange(B)])
    scale = np.concatenate([np.full(nn, 1.0 / (U + C), np.float32), np.ones(B, np.float32)])
    dSr = np.zeros_like(S)
    for i in range(T_):
        ci = (tw[bidx, i] * scale * 2.0 / (D * B))[:, None]
        dSr += ci * (S - P[i])
        np.testing.assert_allclose(dP[i].cpu().numpy(), -ci * (S - P[i]), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(dS.cpu().numpy(), dSr, rtol=1e-4, atol=1e-8)


def test_score_bwd_and_gather():
    B, C, D, R = 6, 5, 256, 100
    rs = np.random.RandomState(0)
    vec, user, dscore = rnd((R, D), 1), rnd((B, D), 2), rnd((B, C), 3)
    cidx = rs.permutation(R)[:B * C].reshape(B, C).astype(np.int32)
    dvec, duser = torch.zeros((R, D), device=DEV), torch.ones((B, D), device=DEV)
    T.call("tnr_score_bwd", dev(vec), dev(cidx), dev(user), dev(dscore), dvec, duser, B, C, D)
    torch.cuda.synchronize()
    want = np.zeros((R, D), np.float32)
    want[cidx.reshape(-1)] = (dscore[:, :, None] * user[:, None, :]).reshape(-1, D)
    np.testing.assert_allclose(dvec.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(duser.cpu().numpy(), 1 + np.einsum("bc,bcd->bd", dscore, vec[cidx]), rtol=1e-4, atol=1e-5)
    tbl = rnd((2, R, D), 4)
    idx = rs.randint(0, R, 33).astype(np.int32)
    out = torch.zeros((2, 40, D), device=DEV)
    T.call("tnr_gather_rows", dev(tbl), R, dev(idx), 33, D, 2, out, 40, 5)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy()[:, 5:38], tbl[:, idx])


# ------------------------------------------------------------------------------------------------ optimiser
def test_amsgrad_matches_reference_golden(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "amsgrad.npz"))
    for i in range(2):
        p = dev(z["p0_%d" % i].reshape(-1).copy())
        m, v, vm = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
        for s in range(3):
            T.call("tnr_amsgrad_step", p, dev(z["g%d_%d" % (s, i)].reshape(-1)), m, v, vm, p.numel(), s + 1, 1e-2, 0.9,
                   0.999, 1e-8, 1.0)
            torch.cuda.synchronize()
            np.testing.assert_allclose(p.cpu().numpy(), z["p%d_%d" % (s + 1, i)].reshape(-1), rtol=1e-5, atol=1e-6)


def test_grad_nonfinite_finds_a_single_bad_value_anywhere():
    """tnr_grad_nonfinite (no reference counterpart: the reference steps its optimiser in fp32, run.py:194-195; this build's fp16
    backward carries a loss scale, and a step whose gradient holds inf / nan must be skipped): a lone inf,
    -inf or nan at the first, last or any other position of a buffer of any length (whole 4 KB trips, a tail, fewer than four
    elements) raises guard[0] to the stamp and counts the skip in guard[1]; finite buffers - large values, denormals, zeros -
    leave both alone; guard[2] (the arrival counter) is back at zero either way."""
    rs = np.random.RandomState(5)
    stamp = 1
    for n in (1, 3, 4, 5, 1023, 4095, 4096, 4097, 8192, 12289, (1 << 20) + 3, 9 * (1 << 20) + 1029):
        base = (rs.standard_normal(n) * 1e3).astype(np.float32)
        base[rs.randint(0, n, 4)] = (3.0e38, -3.0e38, 1e-42, 0.0)
        g = torch.zeros(n + 8, device=DEV)[4:4 + n]                      # 16-byte aligned, not the start of an allocation
        guard = torch.zeros(4, dtype=torch.int32, device=DEV)
        skips = 0
        for pos, val in [(None, 0.0), (0, np.inf), (n - 1, np.nan), (int(rs.randint(0, n)), -np.inf), (n // 2, np.nan), (None, 0.0)]:
            a = base.copy()
            if pos is not None:
                a[pos] = val
            g.copy_(torch.from_numpy(a))
            T.call("tnr_grad_nonfinite", g, n, guard, stamp)
            torch.cuda.synchronize()
            got = guard.cpu().numpy()
            skips += pos is not None
            assert (got[0] == stamp) if pos is not None else (got[0] < stamp), (n, pos, got)     # stamps only grow
            assert got[1] == skips and got[2] == 0, (n, pos, got)
            stamp += 1


def test_grad_nonfinite_in_pieces_equals_one_scan():
    """tnr_grad_nonfinite_scan over disjoint slices + one tnr_grad_nonfinite_commit == tnr_grad_nonfinite over the whole buffer
    (data parallelism scans each gradient bucket as its all-reduce lands): the same guard words whichever slice holds the bad value,
    one counted skip per step however many slices found something, none on a clean gradient."""
    n = 3 * 4096 + 2048 + 64
    cuts = [(0, 4096 + 64), (4096 + 64, 3 * 4096), (3 * 4096, n)]
    g = torch.zeros(n, device=DEV)
    ga, gb = torch.zeros(4, dtype=torch.int32, device=DEV), torch.zeros(4, dtype=torch.int32, device=DEV)
    rs = np.random.RandomState(11)
    for stamp, bad in enumerate([(), (5,), (n - 1,), (4096 + 64, 3 * 4096 + 1), (), (7000,)], start=1):
        a = rs.standard_normal(n).astype(np.float32)
        for pos in bad:
            a[pos] = np.inf
        g.copy_(torch.from_numpy(a))
        T.call("tnr_grad_nonfinite", g, n, ga, stamp)
        for s_, e_ in cuts:
            T.call("tnr_grad_nonfinite_scan", g[s_:e_], e_ - s_, gb, stamp)
        T.call("tnr_grad_nonfinite_commit", gb, stamp)
        torch.cuda.synchronize()
        assert torch.equal(ga, gb), (stamp, ga, gb)
    assert int(ga[1]) == 4


def test_refresh_shadows():
    w1, w2 = rnd((200, 768), 1), rnd((768, 96), 2)
    s1, s2 = dev(w1), dev(w2)
    d1 = torch.zeros((256, 768), device=DEV, dtype=torch.bfloat16)
    d1t = torch.zeros((768, 256), device=DEV, dtype=torch.bfloat16)
    d2t = torch.zeros((96, 800), device=DEV, dtype=torch.bfloat16)
    tiles = [((200 + 31) // 32) * (768 // 32), (768 // 32) * 3]
    desc = torch.tensor([[s1.data_ptr(), 200, 768, d1.data_ptr(), 768, d1t.data_ptr(), 256, 0],
                         [s2.data_ptr(), 768, 96, 0, 0, d2t.data_ptr(), 800, 0]], dtype=torch.int64, device=DEV)
    start = torch.tensor([0, tiles[0], tiles[0] + tiles[1]], dtype=torch.int64, device=DEV)
    T.call("tnr_refresh_shadows", desc, 2, sum(tiles), start)
    torch.cuda.synchronize()
    assert np.array_equal(d1.float().cpu().numpy()[:200], bf(w1)) and (d1[200:] == 0).all()
    assert np.array_equal(d1t.float().cpu().numpy()[:, :200], bf(w1).T)
    assert np.array_equal(d2t.float().cpu().numpy()[:, :768], bf(w2).T)


def test_gemm_colsum_epilogue_and_batched_colsum():
    M, N, K = 700, 512, 128
    A, B = bf(rnd((M, K), 1)), bf(rnd((N, K), 2, 0.1))
    aux = bf(rnd((M, N), 3))
    c = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
    rows = T.query("tnr_gemm_colsum_rows", M)
    part = torch.full((rows, N), 7.0, device=DEV)
    T.call("tnr_gemm_nt_ex", dev(A, torch.bfloat16), K, dev(B, torch.bfloat16), K, c, N, M, N, K, None, None, 0,
           dev(aux, torch.bfloat16), N, T.EPI_MULDGELU | T.EPI_COLSUM, part)
    out = torch.zeros(N, device=DEV)
    T.call("tnr_reduce_rows", part, rows, N, N, out, 0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), c.float().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3)
    X = rnd((3, 900, 256), 4)
    o = torch.zeros((3, 256), device=DEV)
    p2 = torch.zeros(3 * T.query("tnr_colsum_part_elems", 900, 256), device=DEV)
    T.call("tnr_colsum_batched", dev(X), 256, 900 * 256, T.F32, 900, 256, 3, o, p2, 0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(o.cpu().numpy(), X.sum(1), rtol=1e-4, atol=1e-3)


def test_reduce_rows_shapes():
    for rows, n in ((825, 1536), (1792, 1), (33, 200), (3, 70)):
        x = rnd((rows, n), rows)
        o = torch.ones(n, device=DEV)
        T.call("tnr_reduce_rows", dev(x), rows, n, n, o, 1)
        torch.cuda.synchronize()
        np.testing.assert_allclose(o.cpu().numpy(), 1 + x.sum(0), rtol=1e-4, atol=1e-3)


# ------------------------------------------------------------------------------------------------ fp16 build
def hf(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.float16).float().numpy()


def test_f16_variants_gemm_attention_ln():
    """Same sources compiled with IEEE half: exact integer GEMMs (NT + wgrad) and attention / LayerNorm spot checks."""
    rs = np.random.RandomState(5)
    M, N, K = 700, 512, 256
    A = rs.randint(-3, 4, (M, K)).astype(np.float32); A[:, 1] += np.arange(M) % 3
    B = rs.randint(-3, 4, (N, K)).astype(np.float32); B[:, 0] += np.arange(N) % 5
    c = torch.zeros((M, N), device=DEV)
    T.call("tnr_gemm_nt_f16", dev(A, torch.float16), K, dev(B, torch.float16), K, c, N, M, N, K, None, None, 0, None, 0, T.EPI_OUTF32)
    Mp = (M + 63) // 64 * 64
    dY, X = np.zeros((Mp, N), np.float32), np.zeros((Mp, K), np.float32)
    dY[:M] = rs.randint(-2, 3, (M, N)); X[:M] = A
    ws = torch.zeros(T.query("tnr_gemm_tn_ws_elems_f16", N, K, 4), device=DEV)
    dW = torch.zeros((N, K), device=DEV)
    T.call("tnr_gemm_tn_wgrad_f16", dev(dY, torch.float16), N, dev(X, torch.float16), K, dW, K, M, N, K, ws, 4, 0)
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy(), A @ B.T) and np.array_equal(dW.cpu().numpy(), dY.T @ X)
    # attention forward in half precision
    Ns, L, Ah, d = 3, 30, 12, 64
    qkv = hf(rnd((Ns * L, 3 * Ah * d), 1))
    mask = (rs.rand(Ns, L) > 0.3).astype(np.float32); mask[0] = 1
    w = rnd((Ah, 32), 2, 0.5)
    relt, madd = torch.zeros((Ah, 32, 32), device=DEV), dev(np.concatenate([(1 - mask) * -10000.0, np.full((Ns, 2), -1e30)], 1).astype(np.float32))
    T.call("tnr_relpos_table", dev(w), Ah, L, relt)
    ctx = torch.zeros((Ns * L, Ah * d), device=DEV, dtype=torch.float16)
    T.call("tnr_attn_l32_fwd_f16", dev(qkv, torch.float16), madd, relt, ctx, Ns, L, Ah)
    torch.cuda.synchronize()
    want = _attn_ref(qkv, mask, O.relpos_bias_table(w, L), Ns, L, Ah)[4]
    np.testing.assert_allclose(ctx.float().cpu().numpy(), want, rtol=3e-3, atol=3e-3)
    # LayerNorm
    x = hf(rnd((77, 768), 3, 2.0)); g, b = 1 + rnd((768,), 4, 0.1), rnd((768,), 5, 0.1)
    y, st = torch.zeros((77, 768), device=DEV, dtype=torch.float16), torch.zeros((77, 2), device=DEV)
    T.call("tnr_ln_fwd_f16", dev(x, torch.float16), dev(g), dev(b), 1e-12, y, st, 77, 768)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.float().cpu().numpy(), O.layer_norm_fwd(x, g, b, 1e-12)[0], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("dtype,tol_f,tol_b", [("bf16", 2e-2, 3e-2), ("fp16", 2e-3, 4e-3)])
@pytest.mark.parametrize("N,L,A", [(2, 24, 3), (2, 33, 2), (1, 128, 12), (2, 200, 2), (1, 512, 2)])
def test_attention_long_fwd_bwd(N, L, A, dtype, tol_f, tol_b):
    """flash-style tiles + online softmax (stage-1 bodies, L up to 512) against the dense oracle math"""
    rd, td, sfx = BUILDS[dtype]
    d = 64
    Lr = (L + 31) // 32 * 32
    rs = np.random.RandomState(L)
    qkv = rd(rnd((N * L, 3 * A * d), 1, 1.0))
    mask = (rs.rand(N, L) > 0.3).astype(np.float32)
    mask[0, :] = 1
    if N > 1:
        mask[1, L // 2:] = 0
    w = rnd((A, 32), 2, 0.5)
    rel = O.relpos_bias_table(w, L)
    madd = np.full((N, Lr), -1e30, np.float32)
    madd[:, :L] = (1.0 - mask) * -10000.0
    relt = torch.zeros((A, Lr, Lr), device=DEV)
    T.call("tnr_relpos_table", dev(w), A, L, relt)
    torch.cuda.synchronize()
    assert np.array_equal(relt.cpu().numpy()[:, :L, :L], rel)
    ctx = torch.zeros((N * L, A * d), device=DEV, dtype=td)
    lse = torch.zeros((N, A, Lr), device=DEV)
    qd, md = dev(qkv, td), dev(madd)
    T.call("tnr_attn_long_fwd" + sfx, qd, md, relt, ctx, lse, N, L, A)
    torch.cuda.synchronize()
    q, k, v, p, want = _attn_ref(qkv, mask, rel, N, L, A)
    np.testing.assert_allclose(ctx.float().cpu().numpy(), want, rtol=tol_f, atol=tol_f)
    s = q @ k.transpose(0, 1, 3, 2) / 8.0 + ((1.0 - mask) * -10000.0)[:, None, None, :] + rel[None]
    lse_ref = np.log(np.exp(s - s.max(-1, keepdims=True)).sum(-1)) + s.max(-1)
    np.testing.assert_allclose(lse.cpu().numpy()[:, :, :L], lse_ref, rtol=1e-3, atol=tol_f)
    dctx = rd(rnd((N * L, A * d), 3))
    dqkv = torch.zeros((N * L, 3 * A * d), device=DEV, dtype=td)
    delta = torch.zeros((N, A, Lr), device=DEV)
    T.call("tnr_attn_long_bwd" + sfx, qd, md, relt, ctx, dev(dctx, td), lse, delta, dqkv, N, L, A)
    torch.cuda.synchronize()
    dch = dctx.reshape(N, L, A, d).transpose(0, 2, 1, 3)
    dp = dch @ v.transpose(0, 1, 3, 2)
    dv = p.transpose(0, 1, 3, 2) @ dch
    ds = p * (dp - (dp * p).sum(-1, keepdims=True))
    back = lambda t: t.transpose(0, 2, 1, 3).reshape(N * L, A * d)
    want_d = np.concatenate([back(ds @ k / 8.0), back(ds.transpose(0, 1, 3, 2) @ q / 8.0), back(dv)], 1)
    got = dqkv.float().cpu().numpy()
    np.testing.assert_allclose(got, want_d, rtol=tol_b, atol=tol_b * np.abs(want_d).max())


@pytest.mark.parametrize("use_mask", [0, 1])
def test_nrms_self_attention_fwd_bwd(use_mask):
    """tnr_user_blend_fwd/bwd + tnr_nrms_attn_fwd/bwd against the oracle's MultiHeadSelfAttention (model_bert.py:37-100)."""
    nm, B, U, D, NH, R = 2, 3, 50, 256, 16, 200
    rs = np.random.RandomState(3 + use_mask)
    vec = rnd((nm, R, D), 1, 0.5)
    hidx = rs.permutation(R)[:B * U].reshape(B, U).astype(np.int32)
    mask = (rs.rand(B, U) > 0.4).astype(np.float32)
    mask[0] = 1
    pad = rnd((nm, D), 2)
    W = rnd((nm, 3 * D, D), 3, 2.0 / np.sqrt(D))
    bq = rnd((nm, 3 * D), 4, 0.1)
    hv = torch.zeros((nm, B * U, D), device=DEV)
    T.call("tnr_user_blend_fwd", dev(vec), R, dev(hidx), dev(mask), dev(pad), use_mask, hv, nm, B, U, D)
    hv_ref = np.stack([vec[z][hidx] if use_mask else vec[z][hidx] * mask[..., None] + pad[z][None, None] * (1 - mask[..., None])
                       for z in range(nm)], 0).astype(np.float32)
    torch.cuda.synchronize()
    np.testing.assert_allclose(hv.cpu().numpy().reshape(nm, B, U, D), hv_ref, rtol=1e-6, atol=1e-7)
    qkv = np.einsum("zrd,zjd->zrj", hv_ref.reshape(nm, B * U, D), W) + bq[:, None, :]
    ctx_rows = B * U + 7
    ctx = torch.zeros((nm, ctx_rows, D), device=DEV)
    T.call("tnr_nrms_attn_fwd", dev(qkv.astype(np.float32)), dev(mask), use_mask, ctx, ctx_rows, nm, B, U, NH)
    torch.cuda.synchronize()
    caches = []
    for z in range(nm):
        ref, c = O.mhsa_fwd(hv_ref[z], W[z][:D], bq[z][:D], W[z][D:2 * D], bq[z][D:2 * D], W[z][2 * D:], bq[z][2 * D:], NH,
                            mask if use_mask else None)
        caches.append(c)
        np.testing.assert_allclose(ctx[z, :B * U].cpu().numpy().reshape(B, U, D), ref, rtol=2e-4, atol=2e-5)
    assert float(ctx[:, B * U:].abs().max()) == 0.0
    # backward of model 0
    dctx = rnd((B, U, D), 9)
    dx_ref, G = O.mhsa_bwd(dctx, caches[0], W[0][:D], W[0][D:2 * D], W[0][2 * D:])
    dqkv = torch.zeros((B * U, 3 * D), device=DEV)
    T.call("tnr_nrms_attn_bwd", dev(qkv[0].astype(np.float32)), dev(mask), use_mask, dev(dctx.reshape(B * U, D)), dqkv, B, U, NH)
    torch.cuda.synchronize()
    dq = dqkv.cpu().numpy()
    x2 = hv_ref[0].reshape(B * U, D)
    for i, n in enumerate(("W_Q", "W_K", "W_V")):
        blk = dq[:, i * D:(i + 1) * D]
        np.testing.assert_allclose(blk.T @ x2, G[n + ".weight"], rtol=2e-3, atol=2e-3 * np.abs(G[n + ".weight"]).max(), err_msg=n)
        if n != "W_K":     # the key bias is a mathematical no-op
            np.testing.assert_allclose(blk.sum(0), G[n + ".bias"], rtol=2e-3, atol=2e-3 * np.abs(G[n + ".bias"]).max())
    np.testing.assert_allclose(dq @ W[0], dx_ref.reshape(B * U, D), rtol=2e-3, atol=2e-3 * np.abs(dx_ref).max())
    # blend backward
    dvec = torch.zeros((R, D), device=DEV)
    part = torch.zeros((B, D + 5), device=DEV)
    T.call("tnr_user_blend_bwd", dev(dx_ref.reshape(B * U, D)), dev(mask), dev(hidx), use_mask, dvec, part[:, 5:], D + 5, B, U, D)
    torch.cuda.synchronize()
    m = np.ones_like(mask)[..., None] if use_mask else mask[..., None]
    want = np.zeros((R, D), np.float32)
    np.add.at(want, hidx.reshape(-1), (dx_ref * m).reshape(-1, D))
    np.testing.assert_allclose(dvec.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(part[:, 5:].sum(0).cpu().numpy(), (dx_ref * (1 - m)).sum((0, 1)), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mean", [0, 1])
def test_cls_and_mean_pooling(mean):
    n_seq, L, H = 37, 30, 768
    y = rnd((n_seq * L, H), 1)
    yb = dev(y, torch.bfloat16)
    nv = torch.zeros((n_seq, H), device=DEV)
    T.call("tnr_pool_fwd", yb, nv, n_seq, L, H, mean)
    yf = yb.float().cpu().numpy().reshape(n_seq, L, H)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nv.cpu().numpy(), yf.mean(1) if mean else yf[:, 0], rtol=1e-5, atol=1e-6)
    dnv = rnd((n_seq, H), 2)
    dy = torch.full((n_seq * L, H), 7.0, device=DEV, dtype=torch.bfloat16)
    T.call("tnr_pool_bwd", dev(dnv), dy, n_seq, L, H, mean)
    torch.cuda.synchronize()
    want = np.repeat(dnv[:, None, :] / L, L, 1) if mean else np.concatenate([dnv[:, None, :], np.zeros((n_seq, L - 1, H), np.float32)], 1)
    np.testing.assert_allclose(dy.float().cpu().numpy().reshape(n_seq, L, H), want, rtol=1e-2, atol=1e-3)


def test_sgemm_group_equals_separate_calls_bit_for_bit():
    """tnr_sgemm_group: several independent fp32 GEMMs (plain, transposed-A split-K with accumulation, batched with bias, batched
    split-K) in one launch + one grouped split reduce -- every output bit-identical to its own tnr_sgemm call."""
    rs = np.random.RandomState(11)
    r = lambda *shape: dev(rs.standard_normal(shape).astype(np.float32))
    dvec, nv, wd = r(1760, 256), r(1760, 768), r(256, 768)
    dP, X = r(4, 1792, 256), r(4, 1792, 256)
    Wt, bt = r(4, 256, 256), r(4, 256)
    c0 = r(256, 768)
    probs = [
        dict(A=dvec, a_rs=1, a_cs=256, sA=0, B=nv, b_rs=1, b_cs=768, sB=0, C=None, ldc=768, sC=0, bias=None, sBias=0, M=256, N=768, K=1760,
             batch=1, alpha=1.0, beta=1.0, ksplit=8, shape=(256, 768), init=c0),                                  # dW = dY^T X, accumulated
        dict(A=dvec, a_rs=256, a_cs=1, sA=0, B=wd, b_rs=1, b_cs=768, sB=0, C=None, ldc=768, sC=0, bias=None, sBias=0, M=1760, N=768, K=256,
             batch=1, alpha=1024.0, beta=0.0, ksplit=1, shape=(1760, 768), init=None),                            # dX = dY W, scaled
        dict(A=X, a_rs=256, a_cs=1, sA=1792 * 256, B=Wt, b_rs=256, b_cs=1, sB=65536, C=None, ldc=256, sC=1792 * 256, bias=bt, sBias=256,
             M=1792, N=256, K=256, batch=4, alpha=1.0, beta=0.0, ksplit=1, shape=(4, 1792, 256), init=None),      # batched projection + bias
        dict(A=dP, a_rs=1, a_cs=256, sA=1792 * 256, B=X, b_rs=1, b_cs=256, sB=1792 * 256, C=None, ldc=256, sC=65536, bias=None, sBias=0,
             M=256, N=256, K=1792, batch=4, alpha=1.0, beta=0.0, ksplit=8, shape=(4, 256, 256), init=None),       # batched split-K
    ]
    outs = {}
    for mode in ("separate", "group"):
        cs, parts = [], []
        for q in probs:
            c = q["init"].clone() if q["init"] is not None else torch.full(q["shape"], 3.0, device=DEV)
            cs.append(c)
            parts.append(torch.zeros(q["ksplit"] * q["batch"] * q["M"] * q["N"], device=DEV) if q["ksplit"] > 1 else None)
        if mode == "separate":
            for q, c, p_ in zip(probs, cs, parts):
                T.call("tnr_sgemm", q["A"], q["a_rs"], q["a_cs"], q["sA"], None, q["B"], q["b_rs"], q["b_cs"], q["sB"], c, q["ldc"], q["sC"],
                       q["bias"], q["sBias"], q["M"], q["N"], q["K"], q["batch"], q["alpha"], q["beta"], q["ksplit"], p_)
        else:
            T.sgemm_group([dict({k: v for k, v in q.items() if k not in ("shape", "init")}, C=c, part=p_) for q, c, p_ in zip(probs, cs, parts)])
        torch.cuda.synchronize()
        outs[mode] = cs
    for a, b in zip(outs["separate"], outs["group"]):
        assert torch.equal(a, b)
    want = c0.cpu().numpy() + dvec.cpu().numpy().T @ nv.cpu().numpy()
    np.testing.assert_allclose(outs["group"][0].cpu().numpy(), want, rtol=1e-4, atol=2e-3)
