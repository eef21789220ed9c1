"""A handful of NT GEMM launches at the step's shapes, for rocprofv3 --pmc passes (tools/pmc_gemm.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
dev, M, td, sfx = "cuda:0", 52800, torch.float16, "_f16"
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    T.lib().tnr_gemm_set_option(k.encode(), int(v))
for (N, K, fl) in ((3072, 768, 0), (768, 3072, 0), (2304, 768, 1)):
    a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
    c = torch.zeros((M, N), device=dev, dtype=td); bias = torch.randn(N, device=dev)
    for _ in range(6):
        T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, bias, None, 0, None, 0, fl, None)
    torch.cuda.synchronize()
