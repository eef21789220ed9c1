"""Shader clock the chip holds during the ping-pong NT GEMM (tools/_probe build with -DTNR_PROBES=1: two s_memtime /
s_memrealtime stamps per workgroup, nothing else): cycles / (100 MHz ticks) per workgroup, median over workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import numpy as np, torch, tnr_hip as T
T.LIB_PATH = os.path.join(ROOT, "tools", "_probe", "libtnr_hip.so")
dev, M, td, sfx = "cuda:0", 52800, torch.float16, "_f16"
T.lib().tnr_gemm_set_option(b"probe", 64)
for (N, K) in ((3072, 768), (768, 3072), (2304, 768)):
    a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
    c = torch.zeros((M, N), device=dev, dtype=td)
    cs = torch.zeros((T.query("tnr_gemm_colsum_rows", M), N), device=dev)
    def run():
        T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, None, None, 0, None, 0, 0, cs)    # colsum buffer = stamp area (flag not set)
    for _ in range(300): run()               # ~0.1 s of back-to-back launches: let the clock settle
    torch.cuda.synchronize()
    st = cs.view(torch.int64).reshape(-1)[:512].cpu().numpy().reshape(256, 2)
    mhz = st[:, 0] / np.maximum(st[:, 1], 1) * 100.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print("N=%d K=%d: %.1f us (%.0f TF) ; shader clock median %.0f MHz (min %.0f max %.0f) ; workgroup life %.1f us" % (
        N, K, us, 2.0 * M * N * K / us / 1e6, np.median(mhz), mhz.min(), mhz.max(), np.median(st[:, 1]) / 100.0))
