"""Lifetimes of the 256 persistent workgroups of an NT launch (probe 64 of the -DTNR_PROBES=2 build): how much of a launch is the
spread between the first and the last workgroup to run out of tiles?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import numpy as np, torch, tnr_hip as T
T.LIB_PATH = os.path.join(ROOT, "tools", "_probe", "libtnr_hip.so")
dev, td, sfx = "cuda:0", torch.float16, "_f16"
T.lib().tnr_gemm_set_option(b"probe", 64)
for M in (52800, 211200):
    for (N, K) in ((2304, 768), (768, 3072), (768, 768), (3072, 768)):
        a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
        c = torch.zeros((M, N), device=dev, dtype=td)
        cs = torch.zeros((T.query("tnr_gemm_colsum_rows", M), N), device=dev)
        run = lambda: T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, None, None, 0, None, 0, 0, cs)
        for _ in range(30): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        life = cs.view(torch.int64).reshape(-1)[:512].cpu().numpy().reshape(256, 2)[:, 1] / 100.0
        print("M=%6d N=%4d K=%4d: launch %.1f us ; workgroup life min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us ; mean idle behind the last one %.1f us (%.1f %%)" % (
            M, N, K, us, life.min(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.max(), life.max() - life.mean(),
            100 * (life.max() - life.mean()) / us), flush=True)
