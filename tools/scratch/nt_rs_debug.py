"""pp=1 vs pp=2 (register-staged NT kernel) on one launch shape: which 256 x 256 tiles differ; GPU box.
N=2304 K=768 FLAGS=1 M=52800 python tools/scratch/nt_rs_debug.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
dev = "cuda:0"
M, N, K, fl = [int(os.environ.get(k, d)) for k, d in (("M", 52800), ("N", 2304), ("K", 768), ("FLAGS", 1))]
td, sfx = torch.float16, "_f16"
g = torch.Generator(device=dev).manual_seed(3)
a = (torch.randn((M, K), device=dev, generator=g) * 0.5).to(td); b = (torch.randn((N, K), device=dev, generator=g) * 0.05).to(td)
bias = torch.randn(N, device=dev, generator=g); r = torch.randn((M, N), device=dev, generator=g).to(td)
aux0 = torch.randn((M, N), device=dev, generator=g).to(td)
outs = []
for pp in (1, 2):
    T.lib().tnr_gemm_set_option(b"pp", pp)
    c = torch.zeros((M, N), device=dev, dtype=torch.float32 if fl & 32 else td)
    aux = aux0.clone() if fl & (64 | 16) else None
    cs = torch.zeros((T.query("tnr_gemm_colsum_rows", M), N), device=dev) if fl & 128 else None
    T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, bias if fl & 1 else None, r if fl & 8 else None, N if fl & 8 else 0,
           aux, N if aux is not None else 0, fl, cs)
    torch.cuda.synchronize()
    outs.append(c.float())
d = (outs[0] != outs[1])
print("M %d N %d K %d flags %d: %d of %d elements differ" % (M, N, K, fl, int(d.sum()), d.numel()))
if d.any():
    rows = d.any(1).nonzero().flatten(); cols = d.any(0).nonzero().flatten()
    print("rows %d .. %d (%d rows), cols %d .. %d (%d cols)" % (rows.min(), rows.max(), len(rows), cols.min(), cols.max(), len(cols)))
    ct = sorted(set((cols // 256).tolist())); print("column tiles:", ct)
    rt = (rows // 32).unique().tolist(); print("32-row blocks touched: %d, first %s" % (len(rt), rt[:24]))
    i, j = d.nonzero()[0].tolist(); print("first diff at", (i, j), float(outs[0][i, j]), float(outs[1][i, j]))
    zero2 = (outs[1] == 0) & d
    print("of the differing elements, pp=2 wrote 0 in %d" % int(zero2.sum()))
