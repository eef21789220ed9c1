"""Interleaved same-process A/B of TWO BUILDS of the library over the encoder's NT shapes (kernel changes without a runtime switch):
    OLD=tools/_probe/libtnr_old.so python tools/gemm_ab_lib.py          (old = a build of the previous commit's sources)"""
import collections, ctypes, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
def load(tag, path):
    spec = importlib.util.spec_from_file_location("tnr_" + tag, os.path.join(ROOT, "tiny-newsrec_amd", "tnr_hip.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    L = ctypes.CDLL(path)
    for name, args in m._SIG.items():
        if hasattr(L, name):
            fn = getattr(L, name); fn.argtypes = args; fn.restype = m._RET.get(name, m._I)
    L.tnr_last_error.restype = ctypes.c_char_p
    m._lib = L
    return m
libs = [load("old", os.path.join(ROOT, os.environ.get("OLD", "tools/_probe/libtnr_old.so"))),
        load("new", os.path.join(ROOT, "tiny-newsrec_amd", "csrc", "libtnr_hip.so"))]
for kv in filter(None, os.environ.get("OPT", "").split(",")):          # options fixed for BOTH libraries, e.g. OPT=allow_fine=0
    for m_ in libs:
        m_.lib().tnr_gemm_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1]))
dev, M = "cuda:0", int(os.environ.get("M", 52800))
td, sfx = torch.float16, "_f16"
SHAPES = ((3072, 768, 0), (3072, 768, 67), (3072, 768, 3), (3072, 768, 16 | 128), (768, 3072, 9), (2304, 768, 1), (768, 768, 9), (768, 768, 0),
          (768, 2304, 8), (768, 3072, 8), (256, 768, 1 | 4 | 32), (768, 256, 8))
tot = [0.0, 0.0]
if os.environ.get("MODE") == "tn":                 # weight-gradient shapes with the engine's split counts
    sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
    import engine as E
    Mp = (M + 127) // 128 * 128
    for (N, K) in ((3072, 768), (768, 3072), (2304, 768), (768, 768), (256, 768)):
        dy = torch.zeros((Mp, N), device=dev, dtype=td); dy[:M] = (torch.randn((M, N), device=dev) * 0.1).to(td)
        x = torch.zeros((Mp, K), device=dev, dtype=td); x[:M] = torch.randn((M, K), device=dev).to(td)
        dw = torch.zeros((N, K), device=dev)
        sp = E.Engine._wgrad_splits(N, K)[0]
        ws = torch.zeros(N * K * sp, device=dev)
        acc = collections.defaultdict(list)
        for rnd in range(8):
            for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
                run = lambda: libs[v].call("tnr_gemm_tn_wgrad" + sfx, dy, N, x, K, dw, K, M, N, K, ws, sp, 0)
                run(); run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(10): run()
                e1.record(); torch.cuda.synchronize()
                acc[v].append(e0.elapsed_time(e1) * 100)
        m0, m1 = sorted(acc[0])[4], sorted(acc[1])[4]
        tot[0] += m0; tot[1] += m1
        print("dW %4d x %4d splits %2d: old %.1f us (%.0f TF)   new %.1f us (%.0f TF)   (%+.1f %%)" % (
            N, K, sp, m0, 2.0 * M * N * K / m0 / 1e6, m1, 2.0 * M * N * K / m1 / 1e6, 100 * (m1 - m0) / m0), flush=True)
    print("sum: %.1f us vs %.1f us (%+.1f %%)" % (tot[0], tot[1], 100 * (tot[1] - tot[0]) / tot[0]))
    sys.exit(0)
for (N, K, fl) in SHAPES:
    a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
    c = torch.zeros((M, N), device=dev, dtype=torch.float32 if fl & 32 else td)
    bias = torch.randn(N, device=dev); r = torch.randn((M, N), device=dev).to(td); aux = torch.randn((M, N), device=dev).to(td)
    cs = torch.zeros((libs[1].query("tnr_gemm_colsum_rows", M), N), device=dev) if fl & 128 else None
    def run(T):
        T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, bias, r if fl & 8 else None, N if fl & 8 else 0,
               aux if fl & (64 | 16) else None, N if fl & (64 | 16) else 0, fl, cs)
    acc = collections.defaultdict(list)
    for rnd in range(8):
        for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            for _ in range(2): run(libs[v])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run(libs[v])
            e1.record(); torch.cuda.synchronize()
            acc[v].append(e0.elapsed_time(e1) * 100)
    m0, m1 = sorted(acc[0])[4], sorted(acc[1])[4]
    tot[0] += m0; tot[1] += m1
    print("N=%4d K=%4d flags %3d: old %.1f us (%.0f TF)   new %.1f us (%.0f TF)   (%+.1f %%)" % (
        N, K, fl, m0, 2.0 * M * N * K / m0 / 1e6, m1, 2.0 * M * N * K / m1 / 1e6, 100 * (m1 - m0) / m0), flush=True)
print("sum: %.1f us vs %.1f us (%+.1f %%)" % (tot[0], tot[1], 100 * (tot[1] - tot[0]) / tot[0]))
