"""GPU soak of the tile queue (include/tnr_hip.h: tnr_gemm_queue_reset; csrc/gemm.hip pp_q_fetch / pp_q_wait): the persistent GEMM
kernels hand out their tiles through per-stream counters, and the index of a workgroup's next tile sits in a register the compiler
cannot see while the atomic is in flight (tools/check_pending_spill.py guards the build).  Parity tests check tile coverage at a
handful of launches; this one launches every persistent kernel 500 times on three streams AT ONCE (three counter sets, workgroups
of three launches competing for the CUs) with integer operands and an ACCUMULATING epilogue, so that a tile nobody took, a tile
taken twice or a garbage tile index all leave a residue:

  * NT kernels (256- and 224-row tiles): C <- (+-A) B^T + C in place (TNR_EPI_RES with res = C), the sign of A alternating - after
    every pair of launches C is back at its start value exactly; a tile computed twice in a launch adds its product twice, a
    skipped one misses it.
  * weight-gradient kernel (gemm_tn_rs_kernel; its grouped launch with two chained problems as well): dW += (+-dY)^T X - a unit
    that nobody computed leaves the previous launch's slab (opposite sign) in the sum.
Operands are small integers: every product and sum is exact in fp16 / fp32, so "back at the start" is bit-for-bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tnr_hip as T                      # noqa: E402

DEV = "cuda:0"
LAUNCHES = 500
R_ = T.EPI_RES


def _ints(shape, lo, hi, g, td):
    return torch.randint(lo, hi + 1, shape, device=DEV, generator=g).to(td)


@pytest.mark.parametrize("M,route", [(33000, T.ROUTE_256), (52800, None)])
def test_nt_persistent_tiles_are_taken_exactly_once_over_500_launches_on_three_streams(M, route):
    td, sfx, N, K = torch.float16, "_f16", 768, 256
    if route is not None:
        assert T.query("tnr_gemm_nt_route" + sfx, M, N, K, R_) in (T.ROUTE_256, T.ROUTE_224)
    g = torch.Generator(device=DEV).manual_seed(11)
    streams = [torch.cuda.Stream(DEV) for _ in range(3)]
    sets = []
    for s in range(3):
        a = _ints((M, K), -1, 1, g, td)
        b = _ints((N, K), -1, 1, g, td)
        c0 = _ints((M, N), -8, 8, g, td)
        sets.append((a, (-a).contiguous(), b, c0, c0.clone()))
    torch.cuda.synchronize()
    # one launch first: the in-place accumulate itself against an fp32 product (exact on these integers)
    a, na, b, c0, c = sets[0]
    T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, None, c, N, None, 0, R_, None)
    torch.cuda.synchronize()
    want = (a.float() @ b.float().t() + c0.float())
    assert torch.equal(c.float(), want)
    T.call("tnr_gemm_nt_ex" + sfx, na, K, b, K, c, N, M, N, K, None, c, N, None, 0, R_, None)
    torch.cuda.synchronize()
    assert torch.equal(c, c0)
    for i in range(LAUNCHES):
        for st, (a, na, b, c0, c) in zip(streams, sets):
            with torch.cuda.stream(st):
                T.call("tnr_gemm_nt_ex" + sfx, a if i % 2 == 0 else na, K, b, K, c, N, M, N, K, None, c, N, None, 0, R_, None)
    torch.cuda.synchronize()
    for s, (a, na, b, c0, c) in enumerate(sets):
        bad = (c != c0).nonzero()
        assert bad.numel() == 0, "stream %d: %d elements off after %d launches, first at %s (tile row %d, col %d)" % (
            s, bad.shape[0], LAUNCHES, bad[0].tolist(), int(bad[0, 0]) // 32, int(bad[0, 1]) // 256)


@pytest.mark.parametrize("grouped", [False, True])
def test_wgrad_persistent_units_are_taken_exactly_once_over_500_launches_on_three_streams(grouped):
    import engine as E
    td, sfx, M, N, K = torch.float16, "_f16", 20000, 768, 768
    g = torch.Generator(device=DEV).manual_seed(13)
    streams = [torch.cuda.Stream(DEV) for _ in range(3)]
    splits, elems = E.Engine._wgrad_splits(N, K)
    Mp = (M + 127) // 128 * 128
    sets = []
    for s in range(3):
        dy = torch.zeros((Mp, N), device=DEV, dtype=td)
        x = torch.zeros((Mp, K), device=DEV, dtype=td)
        dy[:M] = _ints((M, N), -1, 1, g, td)
        x[:M] = _ints((M, K), -1, 1, g, td)
        dw0 = _ints((N, K), -8, 8, g, torch.float32)
        sets.append((dy, (-dy).contiguous(), x, dw0, dw0.clone(), torch.zeros(elems, device=DEV)))
    def launch(dy_, x, dw, ws):
        if not grouped:
            T.call("tnr_gemm_tn_wgrad_ex" + sfx, dy_, N, x, K, dw, K, M, N, K, ws, splits, 1, 1.0)
            return
        # the same sum as two chained problems of one grouped launch (stage 1's form): rows [0, M1) then rows [M1, M)
        M1 = 12800
        s1 = max(1, splits * M1 // M)
        common = dict(dW=dw, lddw=K, N=N, K=K, out_scale=1.0)
        T.wgrad_group([dict(common, dY=dy_, lddy=N, X=x, ldx=K, M=M1, ws=ws, splits=s1, accumulate=1),
                       dict(common, dY=dy_[M1:], lddy=N, X=x[M1:], ldx=K, M=M - M1, ws=None, splits=splits - s1, accumulate=2)], f16=True)

    dy, ndy, x, dw0, dw, ws = sets[0]
    launch(dy, x, dw, ws)
    torch.cuda.synchronize()
    assert torch.equal(dw, dy[:M].float().t() @ x[:M].float() + dw0)
    launch(ndy, x, dw, ws)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw0)
    for i in range(LAUNCHES):
        for st, (dy, ndy, x, dw0, dw, ws) in zip(streams, sets):
            with torch.cuda.stream(st):
                launch(dy if i % 2 == 0 else ndy, x, dw, ws)
    torch.cuda.synchronize()
    for s, (dy, ndy, x, dw0, dw, ws) in enumerate(sets):
        bad = (dw != dw0).nonzero()
        assert bad.numel() == 0, "stream %d: %d elements off after %d launches, first at %s" % (s, bad.shape[0], LAUNCHES, bad[0].tolist())
