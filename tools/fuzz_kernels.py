"""Randomised shape / flag fuzz of the GEMM entry points against fp32 torch references (development aid; GPU box).
Targets the routing boundaries: M around the tile heights (128 / 224 / 256) and the 'sparse grid' threshold, N = 128 * odd
(256x128 kernel), every epilogue combination each kernel family accepts, both 16-bit builds."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import numpy as np, torch
import tnr_hip as T

dev = "cuda:0"
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
edge_m = [1, 2, 63, 64, 65, 127, 128, 129, 223, 224, 225, 255, 256, 257, 447, 448, 449, 511, 512, 513, 9791, 9792, 9793]
for case in range(n_cases):
    f16 = bool(rs.rand() < 0.3)
    dt = torch.float16 if f16 else torch.bfloat16
    sfx = "_f16" if f16 else ""
    M = int(rs.choice(edge_m)) if rs.rand() < 0.5 else int(rs.randint(1, 12000))
    N = int(rs.choice([128, 256, 384, 768, 1280, 2304]))
    K = int(rs.choice([64, 128, 192, 768, 1024]))
    fl = 0
    if rs.rand() < 0.6: fl |= T.EPI_BIAS
    act = rs.randint(4)
    if act == 1: fl |= T.EPI_GELU
    if act == 2: fl |= T.EPI_TANH
    if act == 3: fl |= T.EPI_MULDGELU
    if rs.rand() < 0.4 and act != 3: fl |= T.EPI_RES
    if rs.rand() < 0.3 and act != 3: fl |= T.EPI_AUXOUT
    outf32 = rs.rand() < 0.2
    if outf32: fl |= T.EPI_OUTF32
    colsum = (not outf32) and M > 128 and rs.rand() < 0.3
    if colsum: fl |= T.EPI_COLSUM
    lda, ldb, ldc = K + 8 * int(rs.randint(0, 3)), K + 8 * int(rs.randint(0, 3)), N + 8 * int(rs.randint(0, 3))
    a = (torch.randn((M, lda), device=dev) * 0.5).to(dt); b = (torch.randn((N, ldb), device=dev) * 0.1).to(dt)
    guard = 5
    c = torch.full((M + guard, ldc), 7.0, device=dev, dtype=torch.float32 if outf32 else dt)
    bias = torch.randn(N, device=dev)
    res = torch.randn((M, N), device=dev).to(dt); aux_in = torch.randn((M, N), device=dev).to(dt)
    aux = aux_in.clone() if (fl & T.EPI_MULDGELU) else torch.full((M + guard, N), 3.0, device=dev, dtype=dt)
    cs_rows = T.query("tnr_gemm_colsum_rows" + sfx, M)
    cs = torch.zeros((cs_rows, N), device=dev)
    T.call("tnr_gemm_nt_ex" + sfx, a, lda, b, ldb, c, ldc, M, N, K, bias if fl & T.EPI_BIAS else None, res if fl & T.EPI_RES else None,
           N if fl & T.EPI_RES else 0, aux if fl & (T.EPI_AUXOUT | T.EPI_MULDGELU) else None, N if fl & (T.EPI_AUXOUT | T.EPI_MULDGELU) else 0,
           fl, cs if colsum else None)
    torch.cuda.synchronize()
    v = a[:, :K].float() @ b[:, :K].float().t()
    if fl & T.EPI_BIAS: v = v + bias
    pre = v.clone()
    if fl & T.EPI_GELU: v = torch.nn.functional.gelu(v)
    if fl & T.EPI_TANH: v = torch.tanh(v)
    if fl & T.EPI_MULDGELU:
        u = aux_in.float()
        v = v * (0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi))
    if fl & T.EPI_RES: v = v + res.float()
    got = c[:M, :N].float()
    tol = 2e-2 if not outf32 else 2e-3
    scale = max(1.0, float(v.abs().max()))
    err = float((got - v).abs().max()) / scale
    ok = err < tol and bool((c[M:] == 7.0).all()) and bool((c[:M, N:] == 7.0).all())
    if fl & T.EPI_AUXOUT:
        ok = ok and float((aux[:M].float() - pre).abs().max()) / max(1.0, float(pre.abs().max())) < 2e-2 and bool((aux[M:] == 3.0).all())
    if colsum:
        want = c[:M, :N].float().sum(0)
        ok = ok and float((cs.sum(0) - want).abs().max()) <= 1e-3 * max(1.0, float(want.abs().max())) + 1e-2
    bad += not ok
    if not ok or case % 25 == 0:
        print("%s nt case %3d %s M=%5d N=%4d K=%4d flags=%3d ld=(%d,%d,%d): err %.2e" % ("ok " if ok else "BAD", case, "f16" if f16 else "bf16", M, N, K, fl, lda, ldb, ldc, err), flush=True)
# weight gradients
for case in range(n_cases // 3):
    f16 = bool(rs.rand() < 0.3); dt = torch.float16 if f16 else torch.bfloat16; sfx = "_f16" if f16 else ""
    M = int(rs.choice(edge_m)) if rs.rand() < 0.5 else int(rs.randint(1, 12000))
    N = int(rs.choice([128, 256, 768, 2304])); K = int(rs.choice([128, 256, 768, 3072]))
    Mp = (M + 63) // 64 * 64
    dy = torch.zeros((Mp, N), device=dev, dtype=dt); dy[:M] = (torch.randn((M, N), device=dev) * 0.5).to(dt)
    x = torch.full((Mp, K), 9.0, device=dev, dtype=dt); x[:M] = (torch.randn((M, K), device=dev) * 0.5).to(dt)     # pad rows of x need not be zero
    splits = int(rs.choice([1, 2, 3, 7, 16]))
    ws = torch.zeros(T.query("tnr_gemm_tn_ws_elems" + sfx, N, K, splits), device=dev)
    acc = int(rs.rand() < 0.3)
    dw = torch.randn((N, K), device=dev); dw0 = dw.clone()
    T.call("tnr_gemm_tn_wgrad" + sfx, dy, N, x, K, dw, K, M, N, K, ws, splits, acc)
    torch.cuda.synchronize()
    want = dy[:M].float().t() @ x[:M].float() + (dw0 if acc else 0)
    err = float((dw - want).abs().max()) / max(1.0, float(want.abs().max()))
    ok = err < 2e-3
    bad += not ok
    if not ok or case % 10 == 0:
        print("%s tn case %3d %s M=%5d N=%4d K=%4d splits=%d acc=%d: err %.2e" % ("ok " if ok else "BAD", case, "f16" if f16 else "bf16", M, N, K, splits, acc, err), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
