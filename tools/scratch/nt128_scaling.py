"""What a 128 x 128 NT launch of the stage-1 step waits for: time against K (slope = a K step, intercept = what a launch pays outside
its loop) and against the number of tiles at one tile per CU or fewer (shared path or per-workgroup chain?); GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
if os.environ.get("LIB"):
    T.LIB_PATH = os.path.join(ROOT, os.environ["LIB"], "libtnr_hip.so")
dev, td, sfx = "cuda:0", torch.float16, "_f16"
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
def timeit(fn, cold):
    ts = []
    for _ in range(15):
        if cold: junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[7]
FORCES = ("ver=1 (128 x 128)", "allow_fine=0 (256 x 256 ping-pong)")
if os.environ.get("FINE_ST"):          # LIB=tools/_st4 (the retired four-stage ring applied): both loops of the 128 x 128 kernel
    FORCES = ("ver=1 fine_st=2", "ver=1 fine_st=4")
if os.environ.get("FINE_NW"):          # four waves of 64 x 64 against eight of 32 x 64 per tile
    FORCES = ("ver=1 fine_nw=4", "ver=1 fine_nw=8")
for force in FORCES:
    if "fine_st" in force:
        T.lib().tnr_gemm_set_option(b"fine_st", int(force[-1]))
    if "fine_nw" in force:
        T.lib().tnr_gemm_set_option(b"fine_nw", int(force[-1]))
    T.lib().tnr_gemm_set_option(b"ver", 1 if force.startswith("ver") else 3)
    T.lib().tnr_gemm_set_option(b"allow_fine", 0 if force.startswith("allow") else 1)
    print(force)
    for M in (9600, 4800, 2400, 1200, 600):
        line = "  M=%4d N=768 (%3d tiles of 128^2):" % (M, -(-M // 128) * 6)
        for K in (256, 768, 1536, 3072):
            a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((768, K), device=dev) * 0.05).to(td)
            c = torch.zeros((M, 768), device=dev, dtype=td); bias = torch.randn(768, device=dev); r = torch.randn((M, 768), device=dev).to(td)
            run = lambda: T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, 768, M, 768, K, bias, r, 768, None, 0, T.EPI_BIAS | T.EPI_RES, None)
            run(); run()
            line += "  K=%4d %5.1f / %5.1f us" % (K, timeit(run, False), timeit(run, True))
        print(line + "   (warm / cold caches)", flush=True)
