"""tnr_sgemm of two builds of the library on the same seeded problems (the step's shapes, ragged M / N / K, K splits, batches, every
operand orientation, unaligned views, beta / alpha / bias): outputs saved for a bitwise comparison; GPU box.
    LIB=tools/_old OUT=/tmp/a.pt python tools/scratch/sgemm_check.py ; OUT=/tmp/b.pt python ... ; python ... cmp /tmp/a.pt /tmp/b.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch
if len(sys.argv) > 1 and sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    print("%d outputs, %d differ" % (len(a), len(bad)), bad[:10])
    sys.exit(1 if bad else 0)
import tnr_hip as T
if os.environ.get("LIB"):
    T.LIB_PATH = os.path.join(ROOT, os.environ["LIB"], "libtnr_hip.so")
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
out = {}
part = torch.zeros(64 << 20 >> 2, device=dev)
def run(tag, A, a_rs, a_cs, sA, B, b_rs, b_cs, sB, M, N, K, batch=1, ksplit=1, bias=None, alpha=1.0, beta=0.0):
    C = rnd(batch, M, N) if beta else torch.full((batch, M, N), 7.0, device=dev)
    T.call("tnr_sgemm", A, a_rs, a_cs, sA, None, B, b_rs, b_cs, sB, C, N, M * N, bias, N if bias is not None else 0, M, N, K, batch, alpha, beta,
           ksplit, part if ksplit > 1 else None)
    torch.cuda.synchronize()
    out[tag] = C.cpu()
for (M, N, K) in ((1760, 256, 768), (1792, 256, 256), (160, 256, 768), (1, 200, 256), (77, 130, 70), (64, 64, 64), (65, 63, 129), (1600, 256, 200), (5, 1, 256), (300, 200, 1000)):
    for bt in (1, 3):
        A, B = rnd(bt, M, K), rnd(bt, N, K)
        At, Bt = A.transpose(1, 2).contiguous(), B.transpose(1, 2).contiguous()      # (bt, K, M): row-fast views
        bias = rnd(bt, N)
        key = "%dx%dx%d b%d " % (M, N, K, bt)
        run(key + "kk", A, K, 1, M * K, B, K, 1, N * K, M, N, K, bt, bias=bias)
        run(key + "rr", At, 1, M, M * K, Bt, 1, N, N * K, M, N, K, bt)
        run(key + "kr", A, K, 1, M * K, Bt, 1, N, N * K, M, N, K, bt, alpha=0.5, beta=1.0)
        run(key + "rk", At, 1, M, M * K, B, K, 1, N * K, M, N, K, bt)
        if K >= 256:
            for ks in (2, 8):
                run(key + "kk split%d" % ks, A, K, 1, M * K, B, K, 1, N * K, M, N, K, bt, ksplit=ks, bias=bias)
                run(key + "rr split%d" % ks, At, 1, M, M * K, Bt, 1, N, N * K, M, N, K, bt, ksplit=ks, beta=1.0)
# unaligned bases and odd leading dimensions (the float4 path must step aside), general strides
M, N, K = 100, 96, 200
buf = rnd(M * (K + 3) + 8)
A1 = buf[1:1 + M * (K + 3)].view(M, K + 3)          # base off by 4 bytes, row stride K + 3
Bm = rnd(N, K)
run("unaligned k-fast", A1, K + 3, 1, 0, Bm, K, 1, 0, M, N, K)
A2 = rnd(K, 2 * M)                                   # element (m, k) at A2[k][2 m]: neither stride is 1
run("general strides", A2, 2, 2 * M, 0, Bm, K, 1, 0, M, N, K)
A3 = rnd(K, M + 1)[:, 1:]                            # row-fast with an odd column stride and an offset base
run("unaligned row-fast", A3, 1, M + 1, 0, Bm, K, 1, 0, M, N, K)
torch.save(out, os.environ.get("OUT", "/tmp/sgemm.pt"))
print(len(out), "problems")
