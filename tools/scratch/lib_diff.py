"""Compare the forward kernels of two builds of libtnr_hip.so on the same inputs (bitwise)."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
def load(tag, path):
    spec = importlib.util.spec_from_file_location("tnr_" + tag, os.path.join(ROOT, "tiny-newsrec_amd", "tnr_hip.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); m.LIB_PATH = path
    # the old library lacks the new symbols: bind only what exists
    import ctypes
    L = ctypes.CDLL(path)
    for name, args in m._SIG.items():
        if hasattr(L, name):
            fn = getattr(L, name); fn.argtypes = args; fn.restype = m._RET.get(name, m._I)
    L.tnr_last_error.restype = ctypes.c_char_p
    m._lib = L
    return m
A = load("old", os.path.join(ROOT, "_wt", "tiny-newsrec_amd", "csrc", "libtnr_hip.so"))
B = load("new", os.path.join(ROOT, "tiny-newsrec_amd", "csrc", "libtnr_hip.so"))
dev = "cuda:0"; td = torch.float16; sfx = "_f16"
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
def cmp(name, fn):
    outs = []
    for m in (A, B):
        outs.append(fn(m))
    same = all(torch.equal(x, y) for x, y in zip(outs[0], outs[1]))
    d = max(float((x.float() - y.float()).abs().max()) for x, y in zip(outs[0], outs[1]))
    print("%-40s %s  max|diff| %.3e" % (name, "identical" if same else "DIFFERENT", d), flush=True)
M, H, I = 3300, 768, 3072
for (N, K, fl) in ((2304, 768, 1), (768, 768, 9), (3072, 768, 67), (3072, 768, 3), (768, 3072, 9), (256, 768, 37)):
    a, b = rnd(M, K).to(td), rnd(N, K, sc=0.05).to(td); bias = rnd(N); res = rnd(M, N).to(td)
    def run(m):
        c = torch.zeros((M, N), device=dev, dtype=torch.float32 if fl & 32 else td); aux = torch.zeros((M, N), device=dev, dtype=td)
        m.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, bias, res if fl & 8 else None, N if fl & 8 else 0, aux if fl & 64 else None, N if fl & 64 else 0, fl, None)
        torch.cuda.synchronize(); return [c, aux]
    cmp("gemm_nt N=%d K=%d flags=%d" % (N, K, fl), run)
nseq, L, Ah = 110, 30, 12
qkv = rnd(nseq * L, 3 * H).to(td); mask = torch.zeros(nseq, 32, device=dev); mask[:, 20:] = -10000.0; rel = rnd(Ah, 32, 32, sc=0.1)
def run(m):
    ctx = torch.zeros((nseq * L, H), device=dev, dtype=td)
    m.call("tnr_attn_l32_fwd" + sfx, qkv, mask, rel, ctx, nseq, L, Ah); torch.cuda.synchronize(); return [ctx]
cmp("attn_l32_fwd", run)
x = rnd(nseq * L, H).to(td); gm, bt = rnd(H), rnd(H)
def run(m):
    y = torch.zeros_like(x); st = torch.zeros((nseq * L, 2), device=dev)
    m.call("tnr_ln_fwd" + sfx, x, gm, bt, 1e-12, y, st, nseq * L, H); torch.cuda.synchronize(); return [y, st]
cmp("ln_fwd", run)
tok = torch.randint(0, 1000, (nseq, 2 * L), device=dev); tok[:, L:] = 1
word, pos, typ = rnd(1000, H, sc=0.02), rnd(512, H, sc=0.02), rnd(2, H, sc=0.02)
def run(m):
    out = torch.zeros((nseq * L, H), device=dev, dtype=td); ma = torch.zeros((nseq, 32), device=dev)
    m.call("tnr_embed_ln_fwd" + sfx, tok, nseq, L, H, word, pos, typ, gm, bt, 1e-12, out, ma); torch.cuda.synchronize(); return [out, ma]
cmp("embed_ln_fwd", run)
