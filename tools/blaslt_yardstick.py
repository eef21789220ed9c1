"""Vendor-library yardstick for the encoder GEMM shapes (development aid, not part of the product path):
times torch.matmul (hipBLASLt / rocBLAS underneath) on the same M, N, K as tnr_gemm_nt / tnr_gemm_tn_wgrad."""
import os, sys
import torch

dev = "cuda:0"
M = int(os.environ.get("M", 52800))
reps = 10


def t(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304), (256, 768)):
    a = (torch.randn((M, K), device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    us = t(lambda: torch.matmul(a, w.t()))
    us_b = t(lambda: torch.nn.functional.linear(a, w, bias))
    print("NT N=%5d K=%5d  matmul %8.1f us %7.1f TF | linear+bias %8.1f us %7.1f TF" %
          (N, K, us, 2.0 * M * N * K / us / 1e6, us_b, 2.0 * M * N * K / us_b / 1e6))
for N, K in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
    dy = torch.randn((M, N), device=dev).to(torch.bfloat16)
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    us = t(lambda: torch.matmul(dy.t(), x))
    print("TN N=%5d K=%5d  matmul %8.1f us %7.1f TF" % (N, K, us, 2.0 * M * N * K / us / 1e6))
