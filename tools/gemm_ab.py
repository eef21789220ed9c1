"""Interleaved same-process A/B of a runtime GEMM switch (here TNR_GEMM_NT) over the encoder shapes; GPU box.
Box-to-box variance on this pool is up to 15 %, so only interleaved same-box comparisons are meaningful."""
import collections, os, sys
sys.path.insert(0, "tiny-newsrec_amd")
import torch, tnr_hip as T
dev = "cuda:0"
M = int(os.environ.get("M", 52800))
VAR, *VALS = os.environ.get("AB", "TNR_GEMM_NT:0:1").split(":")      # e.g. AB=TNR_GEMM_VER:3:7
res = {}
for (N, K, fl) in ((3072, 768, 0), (3072, 768, 67), (3072, 768, 16), (768, 3072, 9), (2304, 768, 1), (768, 768, 9), (768, 2304, 8)):
    a = (torch.randn((M, K), device=dev) * 0.5).to(torch.bfloat16); b = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
    c = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev); r = torch.randn((M, N), device=dev).to(torch.bfloat16); aux = torch.randn((M, N), device=dev).to(torch.bfloat16)
    def run():
        T.call("tnr_gemm_nt", a, K, b, K, c, N, M, N, K, bias, r if fl & 8 else None, N if fl & 8 else 0, aux if fl & (64 | 16) else None, N if fl & (64 | 16) else 0, fl)
    acc = collections.defaultdict(list)
    for rnd in range(8):
        for nt in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            os.environ[VAR] = VALS[nt]
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            acc[nt].append(e0.elapsed_time(e1) * 100)
    m0, m1 = sorted(acc[0])[4], sorted(acc[1])[4]
    print("N=%4d K=%4d flags %3d : %s=%s %.1f us (%.0f TF)   %s=%s %.1f us (%.0f TF)   (%+.1f %%)" % (
        N, K, fl, VAR, VALS[0], m0, 2.0 * M * N * K / m0 / 1e6, VAR, VALS[1], m1, 2.0 * M * N * K / m1 / 1e6, 100 * (m1 - m0) / m0))
