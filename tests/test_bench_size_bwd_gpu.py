"""GPU parity of the NON-GEMM kernels of the backward AT THE HEADLINE STEP'S SIZE, against the numpy oracle.

The row kernels switch block shape with the row count (norm_embed.hip: lnb_rows / cs_rows change at M = 32 768) and the
small-case tests (tests/test_kernels_gpu.py: M <= 1234, N <= 9 sequences) never reach the branch bench.py runs at
B = 32 (M = 52 800 token rows, n_seq = 1 760, A = 12).  Here every such kernel runs at that size AND on both sides of the
32 767 / 32 768 boundary, in both 16-bit builds, called exactly the way engine.py calls it (partials + tnr_reduce_multi),
and one whole B = 32 backward (4-layer student + 4 teachers, the benchmark's inputs) is compared gradient by gradient
with oracle.model_bwd.

Tolerances: a 16-bit output carries one rounding (2^-8 bf16 / 2^-11 fp16, relative); fp32 column sums over M rows of
16-bit-rounded values are compared with the fp64 sum of the SAME rounded values (what the kernel is specified to add up)
to 1e-5 * sqrt(M) relative; gradients of the full step 1.5e-2 relative L2 (the suite's fp16 bound)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tnr_hip as T                      # noqa: E402
from oracle import dropout_oracle as DO  # noqa: E402
from oracle import newsrec_oracle as O   # noqa: E402

DEV = "cuda:0"
M_BENCH, N_BENCH, H, I, A, L, QPAD = 52800, 1760, 768, 3072, 12, 30, 256
TD = {"bf16": torch.bfloat16, "fp16": torch.float16}
EPS = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}
ROWS = [32767, 32768, M_BENCH]            # both sides of the block-shape switch + the benchmark's row count


def _sfx(dtype):
    return "_f16" if dtype == "fp16" else ""


def rnd(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def q16(x, dtype):
    """fp32 numpy -> rounded to the 16-bit type of the build -> (device 16-bit tensor, fp32 numpy of the rounded values)."""
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(DEV).to(TD[dtype])
    return t, t.float().cpu().numpy()


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def assert_16bit(got, want, dtype, what, extra_abs=0.0):
    """|got - want| <= 1 rounding of the 16-bit output (+ extra_abs for fp32 accumulation-order noise)."""
    e = EPS[dtype]
    np.testing.assert_allclose(got, want, rtol=1.5 * e, atol=1.5 * e * 2.0 ** -6 + extra_abs, err_msg=what)


def assert_sum(got, want, M, scale, what):
    """fp32 column sums over M rows vs the fp64 sum: fixed-order tree sums, error ~ eps32 * sqrt(M) * column r.m.s. * sqrt(M)."""
    tol = 2e-6 * M ** 0.5 * scale * M ** 0.5 + 1e-6
    assert np.abs(got.astype(np.float64) - want).max() <= tol, "%s: max err %.3e > %.3e" % (what, np.abs(got - want).max(), tol)


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M", ROWS)
def test_layernorm_fwd_bwd_at_bench_rows(M, dtype):
    """tnr_ln_fwd + tnr_ln_bwd in the engine's form (dgamma = dbeta = dxsum = NULL, partial rows reduced by the caller) and in
    the self-reducing form, on the 128-row-block branch (M >= 32 768) and the 32-row one (M = 32 767)."""
    sf = _sfx(dtype)
    xd, x = q16(rnd((M, H), 1, 2.0) + 0.3, dtype)
    dyd, dy = q16(rnd((M, H), 2), dtype)
    g, b = 1 + rnd((H,), 3, 0.1), rnd((H,), 4, 0.1)
    y = torch.zeros((M, H), device=DEV, dtype=TD[dtype])
    st = torch.zeros((M, 2), device=DEV)
    T.call("tnr_ln_fwd" + sf, xd, dev(g), dev(b), 1e-12, y, st, M, H)
    yref, cache = O.layer_norm_fwd(x, g, b, 1e-12)
    torch.cuda.synchronize()
    assert_16bit(y.float().cpu().numpy(), yref, dtype, "ln_fwd y", extra_abs=2e-6)
    xh, rstd = cache
    np.testing.assert_allclose(st[:, 0].cpu().numpy(), x.mean(-1, dtype=np.float32), rtol=0, atol=2e-6)
    np.testing.assert_allclose(st[:, 1].cpu().numpy(), rstd[:, 0], rtol=2e-5, atol=0)

    nblk = T.query("tnr_ln_bwd_blocks", M)
    assert nblk == -(-M // (128 if M >= 32768 else 32))
    part = torch.full((T.query("tnr_ln_bwd_part_elems", M, H),), 7.0, device=DEV)
    dx = torch.zeros((M, H), device=DEV, dtype=TD[dtype])
    T.call("tnr_ln_bwd" + sf, dyd, xd, st, dev(g), dx, None, None, None, part, M, H)          # engine.py:945
    torch.cuda.synchronize()
    dxr, dgr, dbr = O.layer_norm_bwd(dy, cache, g)
    dxg = dx.float().cpu().numpy()
    assert_16bit(dxg, dxr, dtype, "ln_bwd dx", extra_abs=3e-6)
    p = part[:nblk * 3 * H].view(nblk, 3, H).cpu().numpy().astype(np.float64).sum(0)           # what tnr_reduce_multi adds up
    sc_g = float(np.sqrt((dy.astype(np.float64) ** 2).mean()))
    assert_sum(p[0], (dy.astype(np.float64) * xh).sum(0), M, sc_g, "dgamma partials")
    assert_sum(p[1], dy.astype(np.float64).sum(0), M, sc_g, "dbeta partials")
    assert_sum(p[2], dxg.astype(np.float64).sum(0), M, float(np.sqrt((dxg.astype(np.float64) ** 2).mean())), "dx column sums (bias gradient)")
    np.testing.assert_allclose(p[0], dgr, rtol=0, atol=2e-3 * np.abs(dgr).max())               # ... and against the oracle's
    np.testing.assert_allclose(p[1], dbr, rtol=0, atol=2e-3 * np.abs(dbr).max())
    # the self-reducing form (dgamma / dbeta adjacent, dxsum): same numbers through tnr_reduce_rows
    dgb, dxs = torch.zeros(2 * H, device=DEV), torch.zeros(H, device=DEV)
    dx2 = torch.zeros_like(dx)
    T.call("tnr_ln_bwd" + sf, dyd, xd, st, dev(g), dx2, dgb, dgb[H:], dxs, part, M, H)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2)
    np.testing.assert_allclose(dgb[:H].cpu().numpy(), p[0], rtol=0, atol=1e-4 * np.abs(p[0]).max())
    np.testing.assert_allclose(dgb[H:].cpu().numpy(), p[1], rtol=0, atol=1e-4 * np.abs(p[1]).max() + 1e-5)
    np.testing.assert_allclose(dxs.cpu().numpy(), p[2], rtol=0, atol=1e-4 * np.abs(p[2]).max() + 1e-4)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_layernorm_bwd_with_dropout_at_bench_rows(dtype):
    """tnr_ln_bwd_do (stage-0 / stage-1 train mode: the Linear in front of the LayerNorm was followed by dropout): second output
    dx * mask / (1 - p) with the oracle's mask bits, its column sums in the partials, at M = 52 800."""
    M, sf = M_BENCH, _sfx(dtype)
    p_, seed, layer, call = 0.1, 20240229, 3, 5
    xd, x = q16(rnd((M, H), 5, 1.5), dtype)
    dyd, dy = q16(rnd((M, H), 6), dtype)
    g = 1 + rnd((H,), 7, 0.1)
    st = torch.zeros((M, 2), device=DEV)
    y = torch.zeros((M, H), device=DEV, dtype=TD[dtype])
    T.call("tnr_ln_fwd" + sf, xd, dev(g), dev(np.zeros(H, np.float32)), 1e-12, y, st, M, H)
    nblk = T.query("tnr_ln_bwd_blocks", M)
    part = torch.zeros(T.query("tnr_ln_bwd_part_elems", M, H), device=DEV)
    dx, dxm = torch.zeros((M, H), device=DEV, dtype=TD[dtype]), torch.zeros((M, H), device=DEV, dtype=TD[dtype])
    site = T.Dropout.site_of(p_, seed, T.DROP_FFN_OUT, layer, call)
    T.call("tnr_ln_bwd_do" + sf, dyd, xd, st, dev(g), dx, None, None, None, part, M, H, dxm, site)
    torch.cuda.synchronize()
    _, cache = O.layer_norm_fwd(x, g, np.zeros(H, np.float32), 1e-12)
    dxr, _, _ = O.layer_norm_bwd(dy, cache, g)
    mask = DO.rows_mask(p_, seed, DO.site_id(DO.KIND_FFN_OUT, layer), call, M, H)
    assert_16bit(dx.float().cpu().numpy(), dxr, dtype, "dx (residual branch)", extra_abs=3e-6)
    got_m = dxm.float().cpu().numpy()
    assert (got_m[mask == 0] == 0).all() and 0.08 < (mask == 0).mean() < 0.12          # dropped elements are exact zeros
    assert_16bit(got_m, dxr * mask, dtype, "dx * mask / (1 - p)", extra_abs=4e-6)
    p = part[:nblk * 3 * H].view(nblk, 3, H).cpu().numpy().astype(np.float64).sum(0)
    assert_sum(p[2], got_m.astype(np.float64).sum(0), M, float(np.sqrt((got_m.astype(np.float64) ** 2).mean())), "masked dx column sums")


# ------------------------------------------------------------------------------------------------ column sums
@pytest.mark.parametrize("M", ROWS)
def test_colsum_at_bench_rows(M):
    """tnr_colsum on the 512-row-block branch (M >= 32 768) and the 64-row one, 16-bit (both builds) and fp32 input, with and
    without accumulation, N = 3 H (the q/k/v bias gradient of the long-sequence path) and N = QPAD."""
    for dtype in ("bf16", "fp16"):
        for N in (3 * H, QPAD):
            xd, x = q16(rnd((M, N), N + M % 7, 1.0), dtype)
            out = torch.full((N,), 2.0, device=DEV)
            part = torch.zeros(T.query("tnr_colsum_part_elems", M, N), device=DEV)
            T.call("tnr_colsum" + _sfx(dtype), xd, N, T.BF16, M, N, out, part, 1)
            torch.cuda.synchronize()
            assert_sum(out.cpu().numpy() - 2.0, x.astype(np.float64).sum(0), M, 1.0, "colsum %s N=%d" % (dtype, N))
    xf = rnd((M, QPAD), 11)
    out = torch.zeros(QPAD, device=DEV)
    part = torch.zeros(T.query("tnr_colsum_part_elems", M, QPAD), device=DEV)
    T.call("tnr_colsum", dev(xf), QPAD, T.F32, M, QPAD, out, part, 0)
    torch.cuda.synchronize()
    assert_sum(out.cpu().numpy(), xf.astype(np.float64).sum(0), M, 1.0, "colsum fp32")
    # batched form, as Engine._transform_grads calls it: T x (Rt, D) fp32 -> (T, D)
    T_, Rt, D = 4, N_BENCH + 32, 256
    X = rnd((T_, Rt, D), 12)
    o = torch.zeros((T_, D), device=DEV)
    p2 = torch.zeros(T_ * T.query("tnr_colsum_part_elems", Rt, D), device=DEV)
    T.call("tnr_colsum_batched", dev(X), D, Rt * D, T.F32, Rt, D, T_, o, p2, 0)
    torch.cuda.synchronize()
    assert_sum(o.cpu().numpy(), X.astype(np.float64).sum(1), Rt, 1.0, "colsum batched")


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_attention_l32_fwd_bwd_at_bench_size(dtype):
    """tnr_attn_l32_fwd / _bwd over the benchmark's 1 760 x 12 (sequence, head) pairs, L = 30, incl. the q/k/v bias-gradient
    partial rows (one per sequence) the engine hands to tnr_reduce_multi, padded and all-pad titles among them."""
    N, d, sf = N_BENCH, 64, _sfx(dtype)
    rs = np.random.RandomState(17)
    qd, qkv = q16(rnd((N * L, 3 * A * d), 1, 1.0), dtype)
    ln = np.clip(rs.normal(14, 4, N).astype(np.int64), 3, L)
    mask = (np.arange(L)[None, :] < ln[:, None]).astype(np.float32)
    mask[::97] = 0.0                                         # the all-zero pad news: finite uniform softmax over -10000
    w = rnd((A, 32), 2, 0.5)
    rel = O.relpos_bias_table(w, L)
    madd = np.full((N, 32), -1e30, np.float32)
    madd[:, :L] = (1.0 - mask) * -10000.0
    relt = torch.zeros((A, 32, 32), device=DEV)
    T.call("tnr_relpos_table", dev(w), A, L, relt)
    ctx = torch.zeros((N * L, A * d), device=DEV, dtype=TD[dtype])
    T.call("tnr_attn_l32_fwd" + sf, qd, dev(madd), relt, ctx, N, L, A)
    q, k, v = [qkv[:, i * A * d:(i + 1) * A * d].reshape(N, L, A, d).transpose(0, 2, 1, 3) for i in range(3)]
    s = q @ k.transpose(0, 1, 3, 2) / 8.0 + ((1.0 - mask) * -10000.0)[:, None, None, :] + rel[None]
    p = O.softmax(s.astype(np.float32), -1)
    want = (p @ v).transpose(0, 2, 1, 3).reshape(N * L, A * d)
    torch.cuda.synchronize()
    e = EPS[dtype]
    # P is rounded to 16 bits before P.V (MFMA operand): error <= ~2 roundings of |ctx| <= max|v|
    np.testing.assert_allclose(ctx.float().cpu().numpy(), want, rtol=3 * e, atol=3 * e * np.abs(v).max() * 0.5)
    dcd, dctx = q16(rnd((N * L, A * d), 3), dtype)
    dqkv = torch.zeros((N * L, 3 * A * d), device=DEV, dtype=TD[dtype])
    bpart = torch.full((N, 3 * A * d), 5.0, device=DEV)
    T.call("tnr_attn_l32_bwd" + sf, qd, dev(madd), relt, dcd, dqkv, bpart, N, L, A)
    torch.cuda.synchronize()
    dch = dctx.reshape(N, L, A, d).transpose(0, 2, 1, 3)
    dp = dch @ v.transpose(0, 1, 3, 2)
    dv = p.transpose(0, 1, 3, 2) @ dch
    ds = p * (dp - (dp * p).sum(-1, keepdims=True))
    back = lambda t: t.transpose(0, 2, 1, 3).reshape(N * L, A * d)
    want_d = np.concatenate([back(ds @ k / 8.0), back(ds.transpose(0, 1, 3, 2) @ q / 8.0), back(dv)], 1)
    got = dqkv.float().cpu().numpy()
    # relative L2 per block (q, k, v): the kernel rounds P, dS to 16 bits before their MFMAs
    for i, nm in enumerate("qkv"):
        g_, w_ = got[:, i * A * d:(i + 1) * A * d].astype(np.float64), want_d[:, i * A * d:(i + 1) * A * d].astype(np.float64)
        err = np.sqrt(((g_ - w_) ** 2).sum() / (w_ ** 2).sum())
        assert err < 4 * e, "d%s: relative L2 error %.3e" % (nm, err)
    np.testing.assert_allclose(got, want_d, rtol=0, atol=16 * e * np.abs(want_d).max())
    # bias-gradient partials: row n = column sums of dqkv over sequence n's L tokens (of the ROUNDED values the wgrad sees)
    want_b = got.reshape(N, L, 3 * A * d).astype(np.float64).sum(1)
    np.testing.assert_allclose(bpart.cpu().numpy(), want_b, rtol=0, atol=1e-5 * L * np.abs(got).max() + 1e-6)


# ------------------------------------------------------------------------------------------------ pooling, embedding
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_attpool_fwd_bwd_at_bench_size(dtype):
    """tnr_attpool_fwd / _bwd over 1 760 sequences x 30 tokens: outputs and the three per-sequence partial rows (fc2 weight, fc2
    bias, fc1 bias gradients) against oracle.att_pool_bwd."""
    N, Q, sf = N_BENCH, 200, _sfx(dtype)
    yd, y2 = q16(rnd((N * L, H), 1), dtype)
    y = y2.reshape(N, L, H)
    w1, b1 = rnd((Q, H), 2, 0.05), rnd((Q,), 3, 0.05)
    w2, b2 = rnd((1, Q), 4, 0.2), rnd((1,), 5, 0.05)
    out, c = O.att_pool_fwd(y, w1, b1, w2, b2)
    e = np.zeros((N * L, QPAD), np.float32)
    e[:, :Q] = c["e"].reshape(N * L, Q)
    nv, alpha, den = torch.zeros((N, H), device=DEV), torch.zeros((N, 32), device=DEV), torch.zeros(N, device=DEV)
    ed = dev(e)
    T.call("tnr_attpool_fwd" + sf, yd, ed, QPAD, dev(w2[0]), dev(b2), Q, nv, alpha, den, N, L, H)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nv.cpu().numpy(), out, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(alpha.cpu().numpy()[:, :L], c["w"], rtol=1e-4, atol=1e-6)
    dnv = rnd((N, H), 6)
    dx, g1, gb1, g2, gb2 = O.att_pool_bwd(dnv, c, w1, w2)
    dyd = torch.zeros((N * L, H), device=DEV, dtype=TD[dtype])
    dpre = torch.ones((N * L, QPAD), device=DEV, dtype=TD[dtype])
    dw2p, db2p, db1p = torch.zeros((N, Q), device=DEV), torch.zeros(N, device=DEV), torch.zeros((N, QPAD), device=DEV)
    T.call("tnr_attpool_bwd" + sf, yd, ed, QPAD, dev(w2[0]), Q, dev(dnv), alpha, den, dyd, dpre, QPAD, dw2p, db2p, db1p, N, L, H)
    torch.cuda.synchronize()
    ee = EPS[dtype]
    direct = (c["w"][..., None] * dnv[:, None, :]).reshape(N * L, H)
    np.testing.assert_allclose(dyd.float().cpu().numpy(), direct, rtol=2 * ee, atol=2 * ee * 1e-2)
    dw = (y * dnv[:, None, :]).sum(-1)
    da = c["w"] * (dw - (dw * c["w"]).sum(1, keepdims=True))
    dpre_ref = (da[..., None] * w2[0][None, None, :] * (1 - c["e"] ** 2)).reshape(N * L, Q)
    got = dpre.float().cpu().numpy()
    np.testing.assert_allclose(got[:, :Q], dpre_ref, rtol=2 * ee, atol=2 * ee * np.abs(dpre_ref).max())
    assert (got[:, Q:] == 0).all()
    sc = np.abs(gb1).max()
    np.testing.assert_allclose(db1p.sum(0).cpu().numpy()[:Q], got[:, :Q].astype(np.float64).sum(0), rtol=0, atol=1e-4 * sc + 1e-6)
    np.testing.assert_allclose(db1p.sum(0).cpu().numpy()[:Q], gb1, rtol=0, atol=2 * ee * sc * 4)
    np.testing.assert_allclose(dw2p.sum(0).cpu().numpy(), g2[0], rtol=0, atol=1e-3 * np.abs(g2).max())
    assert abs(float(db2p.sum()) - float(gb2[0])) <= 1e-3 * max(1e-3, np.abs(da).sum())


def test_embed_ln_indexed_at_bench_size():
    """tnr_embed_ln_fwd_indexed: 1 760 sequences gathered by news index from a resident token table (repeats, the pad row 0 and
    the table's last row among them) -> embeddings + LayerNorm + additive mask, both builds, against oracle.embeddings_fwd."""
    N, V, n_news = N_BENCH, 30522, 5000
    rs = np.random.RandomState(23)
    ids = rs.randint(1000, V, (n_news, L))
    ln = np.clip(rs.normal(14, 4, n_news).astype(np.int64), 3, L)
    mk = (np.arange(L)[None, :] < ln[:, None]).astype(np.int64)
    ids = ids * mk
    ids[0], mk[0] = 0, 0
    table = np.concatenate([ids, mk], 1).astype(np.int32)
    nidx = rs.randint(0, n_news, N).astype(np.int32)
    nidx[:7] = [0, 0, n_news - 1, 5, 5, 5, n_news - 1]
    P = {O.BERT + "embeddings.word_embeddings.weight": rnd((V, H), 1, 0.05),
         O.BERT + "embeddings.position_embeddings.weight": rnd((512, H), 2, 0.05),
         O.BERT + "embeddings.token_type_embeddings.weight": rnd((2, H), 3, 0.05),
         O.BERT + "embeddings.LayerNorm.weight": 1 + rnd((H,), 4, 0.1),
         O.BERT + "embeddings.LayerNorm.bias": rnd((H,), 5, 0.1)}
    want = O.embeddings_fwd(P, ids[nidx]).reshape(N * L, H)
    dP = [dev(P[k]) for k in P]
    for dtype in ("bf16", "fp16"):
        out = torch.zeros((N * L, H), device=DEV, dtype=TD[dtype])
        madd = torch.zeros((N, 32), device=DEV)
        T.call("tnr_embed_ln_fwd_indexed" + _sfx(dtype), dev(table), dev(nidx), N, L, H, dP[0], dP[1], dP[2][0], dP[3], dP[4], 1e-12,
               out, madd)
        torch.cuda.synchronize()
        assert_16bit(out.float().cpu().numpy(), want, dtype, "embed_ln " + dtype, extra_abs=3e-6)
        m = madd.cpu().numpy()
        assert np.array_equal(m[:, :L], (1.0 - mk[nidx].astype(np.float32)) * -10000.0) and (m[:, L:] <= -1e29).all()


# ------------------------------------------------------------------------------------------------ partial reductions
def test_reduce_multi_with_the_headline_steps_job_lists():
    """engine._ReduceBatch (two-level tables for tnr_reduce_multi) on the partial buffers of one trainable layer at B = 32:
    LayerNorm partials (413 rows of 3 H), the dgrad epilogue's column-sum rows (828 x I), the attention backward's per-sequence
    rows (1 760 x 3 H), pooling partials; with accumulation and the 1 / loss-scale factor in the descriptor."""
    import engine as E
    M, N = M_BENCH, N_BENCH
    nblk = T.query("tnr_ln_bwd_blocks", M)
    csr = T.query("tnr_gemm_colsum_rows", M)
    assert (nblk, csr) == (413, 828)
    rs = np.random.RandomState(5)
    jobs = [(nblk, 3 * H, 2 * H, 0, 0, 1.0 / 1024), (nblk, 3 * H, H, 2 * H, 1, 1.0 / 1024), (csr, I, I, 0, 0, 1.0 / 1024),
            (N, 3 * H, 3 * H, 0, 1, 1.0), (N, 200, 200, 0, 0, 0.5), (N, 1, 1, 0, 0, 1.0), (32, 713, 713, 0, 0, 1.0)]
    rb = E._ReduceBatch(torch.device(DEV))
    bufs = []
    for rows, stride, n, off, acc, scale in jobs:
        part = (rs.standard_normal((rows, stride)) * 3).astype(np.float32)
        out0 = rs.standard_normal(n).astype(np.float32)
        pd, od = dev(part), dev(out0.copy())
        rb.add(pd[:, off:] if off else pd, rows, stride, n, od, acc, scale)
        bufs.append((part, out0, pd, od, off, n, acc, scale))
    rb.flush()
    torch.cuda.synchronize()
    for part, out0, pd, od, off, n, acc, scale in bufs:
        want = part[:, off:off + n].astype(np.float64).sum(0) * scale + (out0 if acc else 0.0)
        tol = 3e-6 * part.shape[0] ** 0.5 * 3 * part.shape[0] ** 0.5 * scale + 1e-6
        assert np.abs(od.cpu().numpy() - want).max() <= tol, (part.shape, n, np.abs(od.cpu().numpy() - want).max(), tol)
    # replay (the table is built once and reused every step): the in-place first level must have left nothing behind that changes it
    for part, out0, pd, od, off, n, acc, scale in bufs:
        pd.copy_(dev(part)); od.copy_(dev(out0))
    rb.flush()
    torch.cuda.synchronize()
    for part, out0, pd, od, off, n, acc, scale in bufs:
        want = part[:, off:off + n].astype(np.float64).sum(0) * scale + (out0 if acc else 0.0)
        tol = 3e-6 * part.shape[0] ** 0.5 * 3 * part.shape[0] ** 0.5 * scale + 1e-6
        assert np.abs(od.cpu().numpy() - want).max() <= tol


# ------------------------------------------------------------------------------------------------ the whole backward at B = 32
def test_b32_full_backward_matches_oracle():
    """The benchmark's own step (B = 32, 4-layer student training layers 2-3 + 4 teachers, synthetic MIND-shaped inputs, fp16):
    EVERY parameter gradient against oracle.model_bwd on the same inputs at the suite's fp16 bound (1.5e-2 relative L2) -- the
    large-M branches of ln_bwd / colsum, attention backward over 1 760 x 12 pairs, pooling backward, every partial reduction
    and the multi-round weight-gradient GEMMs in their real call chain.  Forward losses ride along (1e-3)."""
    import engine as E
    import hashinit
    import synth
    from schema import FULL, state_shapes
    nl, T_, B, n_news = 4, 4, 32, 4000
    cfg = E.EngineConfig(n_layers=nl, trainable_layers=(2, 3), num_teachers=T_)
    eng = E.Engine(cfg, DEV, max_batch=B, dtype="fp16")
    P = hashinit.init_state_dict(1234, state_shapes(FULL, nl, cfg.D, T_))
    eng.load_state_dict(P)
    comb = synth.news_table(1234, n_news, cfg.L)
    tabs = synth.teacher_tables(1234, T_, n_news, cfg.D)
    hidx, mask, cidx, label = synth.impressions(1235, B, n_news, cfg.U, cfg.C)
    losses, score = eng.forward_indexed(dev(comb), dev(hidx), dev(mask), dev(cidx), dev(label), dev(tabs))
    eng.backward()
    torch.cuda.synchronize()
    ocfg = dict(n_layers=nl, heads=12, trainable_layers=[2, 3], user_log_mask=False, temperature=1.0, coef=0.2)
    c64 = comb.astype(np.int64)
    out = O.model_fwd(P, ocfg, c64[hidx], mask, c64[cidx], label, [tabs[i][hidx] for i in range(T_)],
                      [tabs[i][cidx] for i in range(T_)], keep=True)
    l = losses.cpu().numpy()
    for got, key in ((l[0], "distill_loss"), (l[1], "target_loss"), (l[2], "emb_loss")):
        assert abs(got - float(out[key])) <= 1e-3 * max(1.0, abs(float(out[key]))), (key, got, float(out[key]))
    G = O.model_bwd(P, ocfg, out)
    del out
    nrm = lambda a: float(np.sqrt((a.astype(np.float64) ** 2).sum()))
    worst, worst_k = 0.0, None
    assert set(eng.grads) == set(G), set(eng.grads) ^ set(G)
    for k in eng.grads:
        got, ref = eng.grad(k).cpu().numpy(), G[k]
        if k.endswith("self.key.bias") or k.endswith("att_fc2.bias"):
            assert np.abs(got).max() < 1e-3 and np.abs(ref).max() < 1e-3, k      # mathematical no-ops: rounding noise on both sides
            continue
        err = nrm(got - ref) / (nrm(ref) + 1e-30)
        if err > worst:
            worst, worst_k = err, k
        assert err < 1.5e-2, "%s: relative L2 error %.3e (|ref| %.3e)" % (k, err, nrm(ref))
    print("\nB=32 backward (fp16): worst gradient relative L2 error %.3e (%s)" % (worst, worst_k))
