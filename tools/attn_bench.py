"""Timing of the L <= 32 attention kernels at the headline step's shape (1760 sequences x 12 heads, L = 30); GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
if os.environ.get("LIB"):
    T.LIB_PATH = os.environ["LIB"]
dev, N, L, A, H = "cuda:0", 1760, 30, 12, 768
F16 = os.environ.get("DTYPE", "fp16") == "fp16"
td, sfx = (torch.float16, "_f16") if F16 else (torch.bfloat16, "")
qkv = (torch.randn(N * L, 3 * H, device=dev) * 0.5).to(td)
mask = torch.zeros(N, 32, device=dev); mask[:, 24:30] = -10000.0; mask[:, 30:] = -1e30
rel = torch.randn(A, 32, 32, device=dev) * 0.1
ctx = torch.zeros(N * L, H, device=dev, dtype=td); dctx = (torch.randn(N * L, H, device=dev) * 0.1).to(td)
dqkv = torch.zeros_like(qkv); bp = torch.zeros(N, 3 * H, device=dev)
# something large between launches so that the inputs are not L2-resident from the previous repetition
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def timeit(fn, reps=20):
    ts = []
    for _ in range(reps):
        junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
f = timeit(lambda: T.call("tnr_attn_l32_fwd" + sfx, qkv, mask, rel, ctx, N, L, A))
b = timeit(lambda: T.call("tnr_attn_l32_bwd" + sfx, qkv, mask, rel, dctx, dqkv, bp, N, L, A))
print("attn_l32 fwd %.1f us (%.0f GB/s of 324 MB)   bwd %.1f us (%.0f GB/s of 567 MB)" % (f, 324e6 / f / 1e3, b, 567e6 / b / 1e3))
# the long-sequence kernels (stage-1 bodies): ~53 k tokens per launch as well
for (Ll, Nl) in ((128, 416), (512, 104)):
    q2 = (torch.randn(Nl * Ll, 3 * H, device=dev) * 0.5).to(td)
    m2 = torch.zeros(Nl, Ll, device=dev); m2[:, Ll - 9:] = -10000.0
    r2 = torch.randn(A, Ll, Ll, device=dev) * 0.1
    c2 = torch.zeros(Nl * Ll, H, device=dev, dtype=td); dc2 = (torch.randn(Nl * Ll, H, device=dev) * 0.1).to(td)
    lse = torch.zeros(Nl, A, Ll, device=dev); delta = torch.zeros(Nl, A, Ll, device=dev); dq2 = torch.zeros_like(q2)
    f = timeit(lambda: T.call("tnr_attn_long_fwd" + sfx, q2, m2, r2, c2, lse, Nl, Ll, A), 10)
    b = timeit(lambda: T.call("tnr_attn_long_bwd" + sfx, q2, m2, r2, c2, dc2, lse, delta, dq2, Nl, Ll, A), 10)
    print("attn_long L=%d N=%d: fwd %.1f us   bwd (dq + dkv passes) %.1f us" % (Ll, Nl, f, b))
