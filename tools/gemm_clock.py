"""Shader clock the chip holds during the persistent ping-pong NT GEMM, read with the product library's own measurement hook
(tnr_gemm_clock_stamps, include/tnr_hip.h: two s_memtime / s_memrealtime stamps per workgroup - round 6; rounds 3-5 needed a
-DTNR_PROBES build for it): shader cycles / (100 MHz ticks) per workgroup, median over workgroups.
    python tools/gemm_clock.py [warm=300]      per shape: time per launch, TFLOP/s, clock after `warm` back-to-back launches, and the
                                               clock of single launches separated by 2 ms of idle (what a step's GEMM sees after an
                                               HBM-bound kernel)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import numpy as np, torch, tnr_hip as T
kw = dict(a.split("=") for a in sys.argv[1:])
warm = int(kw.get("warm", 300))
dev, M, td, sfx = "cuda:0", 52800, torch.float16, "_f16"
stamps = torch.zeros((256, 2), device=dev, dtype=torch.int64)


def mhz():
    st = stamps.cpu().numpy()
    ok = st[:, 1] > 0
    return 100.0 * st[ok, 0] / st[ok, 1], np.median(st[ok, 1]) / 100.0


for (N, K) in ((768, 768), (3072, 768), (768, 3072), (2304, 768)):
    a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
    c = torch.zeros((M, N), device=dev, dtype=td)
    def run():
        T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, None, None, 0, None, 0, 0, None)
    T.lib().tnr_gemm_clock_stamps(stamps.data_ptr(), 256)
    torch.cuda.synchronize(); time.sleep(0.05)
    singles = []
    for _ in range(8):                          # cold single launches
        run(); torch.cuda.synchronize()
        singles.append(float(np.median(mhz()[0])))
        time.sleep(0.002)
    for _ in range(warm): run()               # back-to-back: let the clock settle
    torch.cuda.synchronize()
    m, life = mhz()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    T.lib().tnr_gemm_clock_stamps(None, 0)
    us = e0.elapsed_time(e1) * 50
    print("N=%d K=%d: %.1f us (%.0f TF) ; shader clock after %d back-to-back launches: median %.0f MHz (min %.0f max %.0f) ; workgroup life %.1f us ; "
          "single launches 2 ms apart: %s MHz" % (N, K, us, 2.0 * M * N * K / us / 1e6, warm, np.median(m), m.min(), m.max(), life,
                                                   " ".join("%.0f" % x for x in singles)), flush=True)
