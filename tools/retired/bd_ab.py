"""Interleaved same-process A/B of the persistent NT GEMM with the weight operand staged through LDS (option bd = 0) and read
straight from L2 into registers from the preshuffled image (bd = 1), over the encoder's NT launches at M = 52 800; checks that
the two give bit-identical results.   python tools/bd_ab.py   (LIB=<build> for another library, PROBE=n on a probe build)"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
if os.environ.get("LIB"): T.LIB_PATH = os.path.join(ROOT, os.environ["LIB"])
dev, M = "cuda:0", int(os.environ.get("M", 52800))
td, sfx = torch.float16, "_f16"
opt = lambda k, v: T.lib().tnr_gemm_set_option(k.encode(), int(v))
probe = int(os.environ.get("PROBE", 0))
SHAPES = ((3072, 768, 0), (3072, 768, 67), (3072, 768, 3), (3072, 768, 16 | 128), (768, 3072, 9), (2304, 768, 1), (768, 768, 9), (768, 768, 0),
          (768, 2304, 8), (768, 3072, 8), (256, 768, 1 | 4 | 32), (768, 256, 8))
# launches per training step of each shape (4-layer student, train 2-3): the weighted sum below
W = (0, 2, 2, 2, 4, 4, 4, 1, 1, 2, 1, 1)
tot = [0.0, 0.0]; wt = [0.0, 0.0]
for (N, K, fl), wgt in zip(SHAPES, W):
    a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
    bp = torch.empty_like(b)
    T.call("tnr_gemm_preshuffle_b" + sfx, b, K, N, K, bp)
    c = [torch.zeros((M, N), device=dev, dtype=torch.float32 if fl & 32 else td) for _ in range(2)]
    bias = torch.randn(N, device=dev); r = torch.randn((M, N), device=dev).to(td)
    aux = [torch.randn((M, N), device=dev).to(td) for _ in range(2)]; aux[1].copy_(aux[0])
    cs = [torch.zeros((T.query("tnr_gemm_colsum_rows", M), N), device=dev) if fl & 128 else None for _ in range(2)]
    def run(v):
        opt("bd", v)
        T.call("tnr_gemm_nt_ex" + sfx, a, K, bp if v else b, K, c[v], N, M, N, K, bias, r if fl & 8 else None, N if fl & 8 else 0,
               aux[v] if fl & (64 | 16) else None, N if fl & (64 | 16) else 0, fl, cs[v])
    run(0); run(1); torch.cuda.synchronize()
    same = torch.equal(c[0], c[1]) and (not fl & 64 or torch.equal(aux[0], aux[1])) and (cs[0] is None or torch.equal(cs[0], cs[1]))
    if probe: opt("probe", probe)
    acc = collections.defaultdict(list)
    for rnd in range(8):
        for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            for _ in range(2): run(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run(v)
            e1.record(); torch.cuda.synchronize()
            acc[v].append(e0.elapsed_time(e1) * 100)
    opt("probe", 0)
    m0, m1 = sorted(acc[0])[4], sorted(acc[1])[4]
    tot[0] += m0; tot[1] += m1; wt[0] += wgt * m0; wt[1] += wgt * m1
    print("N=%4d K=%4d flags %3d: LDS-staged B %.1f us (%.0f TF)   B direct %.1f us (%.0f TF)   (%+.1f %%)   %s" % (
        N, K, fl, m0, 2.0 * M * N * K / m0 / 1e6, m1, 2.0 * M * N * K / m1 / 1e6, 100 * (m1 - m0) / m0,
        "bit-identical" if same else "RESULTS DIFFER"), flush=True)
opt("bd", 0)
print("sum: %.1f us vs %.1f us (%+.1f %%)" % (tot[0], tot[1], 100 * (tot[1] - tot[0]) / tot[0]))
print("the step's 25 launches: %.1f us vs %.1f us (%+.1f %%)" % (wt[0], wt[1], 100 * (wt[1] - wt[0]) / wt[0]))
