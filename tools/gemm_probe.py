"""Where does a K step of the ping-pong NT GEMM go?  Timing probes of the -DTNR_PROBES library copy (tools/_probe, built by
`tools/probes/build.sh _probe -DTNR_PROBES=2`); the product library has none of this code.
  probe 1: no staging loads after the first two K tiles (MFMA + fragment reads + barriers only)
  probe 2: no fragment reads / MFMAs (the LDS-DMA pipeline + barriers only)
  probe 4: every row tile reads A rows 0-255 (A resident in L2: the load pipeline at L2-hit rates)
  probe 8: no epilogue ; 16: epilogue without its stores ; 32: every tile stores to tile (0, 0) (the stores stay in L2)
Outputs are wrong by construction; only the times mean anything."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
T.LIB_PATH = os.path.join(ROOT, "tools", "_probe", "libtnr_hip.so")
dev, M = "cuda:0", int(os.environ.get("M", 52800))
td, sfx = torch.float16, "_f16"
opt = lambda k, v: T.lib().tnr_gemm_set_option(k.encode(), int(v))
SHAPES = ((3072, 768, 0), (768, 3072, 0))
PROBES = [(0, "full"), (8, "no epilogue"), (1 | 8, "compute only"), (2 | 8, "loads only"),
          (1 | 8 | 512, "fragment reads only (no DMA, no MFMA)"), (1 | 8 | 1024, "MFMA only (no DMA, no reads)"), (8 | 512, "DMA + reads, no MFMA"),
          (8 | 1024, "DMA + MFMA, no reads"), (1 | 2 | 8, "barriers only")]
for (N, K, fl) in SHAPES:
    a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
    c = torch.zeros((M, N), device=dev, dtype=td); bias = torch.randn(N, device=dev); r = torch.randn((M, N), device=dev).to(td)
    def run():
        T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, bias, r if fl & 8 else None, N if fl & 8 else 0, None, 0, fl, None)
    route = T.query("tnr_gemm_nt_route" + sfx, M, N, K, fl)
    bmh = 224 if route == 224 else 256
    tiles = -(-M // bmh) * (N // 256)
    rounds = -(-tiles // 256)
    res = {}
    for rep in range(3):
        for p, name in PROBES:
            opt("probe", p)
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) * 100)
    opt("probe", 64 | 8)
    for _ in range(20): run()
    torch.cuda.synchronize()
    ck = c.view(torch.int64).reshape(-1)[:2].cpu().numpy()
    opt("probe", 0)
    print("   shader clock during the K loops (no epilogue): %.0f MHz" % (ck[0] / max(ck[1], 1) * 100.0))
    print("N=%d K=%d flags %d: %d tiles of %d rows = %.2f rounds" % (N, K, fl, tiles, bmh, tiles / 256.0))
    for p, name in PROBES:
        us = sorted(res[name])[1]
        print("   %-24s %7.1f us   per round %6.2f us   per K step %5.2f us" % (name, us, us / rounds, us / rounds / (K // 64)), flush=True)
