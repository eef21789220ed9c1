"""Interleaved, repeated timing of tnr_gemm_nt epilogue variants on one shape (development aid; GPU box)."""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tiny-newsrec_amd"))
import torch, tnr_hip as T
dev = "cuda:0"
M = int(os.environ.get("M", 52800)); N = int(os.environ.get("N", 3072)); K = int(os.environ.get("K", 768))
a = (torch.randn((M, K), device=dev) * 0.5).to(torch.bfloat16); b = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
c = torch.zeros((M, N), device=dev, dtype=torch.bfloat16); c32 = torch.zeros((M, N), device=dev)
bias = torch.randn(N, device=dev); res = torch.randn((M, N), device=dev).to(torch.bfloat16); aux = torch.randn((M, N), device=dev).to(torch.bfloat16)
def run(fl):
    T.call("tnr_gemm_nt", a, K, b, K, (c32 if fl & 32 else c), N, M, N, K, bias, res if fl & 8 else None, N if fl & 8 else 0,
           aux if fl & (64 | 16) else None, N if fl & (64 | 16) else 0, fl)
flags = [int(x) for x in os.environ.get("FLAGS", "0,1,3,67,16,9").split(",")]
acc = collections.defaultdict(list)
for rnd in range(6):
    for fl in (flags if rnd % 2 == 0 else flags[::-1]):
        for _ in range(2): run(fl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): run(fl)
        e1.record(); torch.cuda.synchronize()
        acc[fl].append(e0.elapsed_time(e1) * 100)
for fl in flags:
    v = sorted(acc[fl]); print("N=%d K=%d flags %3d: median %.1f us (%.0f TF)  min %.1f max %.1f" % (N, K, fl, v[len(v) // 2], 2.0 * M * N * K / v[len(v) // 2] / 1e6, v[0], v[-1]))
