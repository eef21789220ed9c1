"""Where does a wave of the L <= 32 attention forward spend its life?  s_memtime stamps of a -DTNR_ATTN_STAMPS library copy
(tools/_probe): t0 start, t1 = scores available (operand loads + 4 MFMAs), t2 = probabilities, t3 = P.V done, t4 = end."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import numpy as np, torch, tnr_hip as T
T.LIB_PATH = os.path.join(ROOT, "tools", "_probe", "libtnr_hip.so")
dev, N, L, A, H = "cuda:0", 1760, 30, 12, 768
td, sfx = torch.float16, "_f16"
qkv = (torch.randn(N * L, 3 * H, device=dev) * 0.5).to(td)
mask = torch.zeros(N, 32, device=dev); mask[:, 30:] = -1e30
rel = torch.zeros(12 * 1024 + 4096 * 16, device=dev)          # table + stamp area behind it
ctx = torch.zeros(N * L, H, device=dev, dtype=td)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for rep in range(3):
    junk.zero_()
    T.call("tnr_attn_l32_fwd" + sfx, qkv, mask, rel, ctx, N, L, A)
    torch.cuda.synchronize()
st = rel[12 * 1024:].view(torch.int64).reshape(4096, 8)[:, :5].cpu().numpy().astype(np.float64)
d = np.diff(st, axis=1)
print("cycles per wave (median over 4096 sampled waves): loads+QK %d | softmax %d | PV %d | dump+store %d | total %d" % tuple(
    [np.median(d[:, i]) for i in range(4)] + [np.median(st[:, 4] - st[:, 0])]))
print("kernel span in cycles (first start .. last end of the sample): %d" % (st[:, 4].max() - st[:, 0].min()))
order = np.argsort(st[:, 0]); print("wave start times, deciles:", np.percentile(st[:, 0] - st[:, 0].min(), [0, 10, 50, 90, 100]).astype(int))
