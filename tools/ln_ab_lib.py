"""Interleaved same-process A/B of TWO BUILDS of the library on the LayerNorm kernels at the bench shape (M = 52 800, H = 768):
    OLD=tools/_probe/libtnr_old.so python tools/ln_ab_lib.py"""
import ctypes, importlib.util, os, sys
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
dev, M, H = "cuda:0", int(os.environ.get("M", 52800)), 768
td, sfx = torch.float16, "_f16"
x = torch.randn((M, H), device=dev).to(td); dy = (torch.randn((M, H), device=dev) * 0.1).to(td)
gamma = torch.rand(H, device=dev) + 0.5; beta = torch.randn(H, device=dev)
y = [torch.empty_like(x) for _ in libs]; dx = [torch.empty_like(x) for _ in libs]
stats = torch.empty((M, 2), device=dev)
part = [torch.zeros(int(libs[0].query("tnr_ln_bwd_part_elems", M, H)), device=dev) for _ in libs]
for name in ("fwd", "bwd"):
    res = {0: [], 1: []}
    for rnd in range(8):
        for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            if name == "fwd":
                run = lambda: libs[v].call("tnr_ln_fwd" + sfx, x, gamma, beta, 1e-12, y[v], stats, M, H)
            else:
                run = lambda: libs[v].call("tnr_ln_bwd" + sfx, dy, x, stats, gamma, dx[v], None, None, None, part[v], M, H)
            run(); run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) * 100)
    m0, m1 = sorted(res[0])[4], sorted(res[1])[4]
    same = torch.equal(y[0], y[1]) if name == "fwd" else (torch.equal(dx[0], dx[1]) and torch.equal(part[0], part[1]))
    print("ln_%s: old %.1f us   new %.1f us   (%+.1f %%)   results bit-identical: %s" % (name, m0, m1, 100 * (m1 - m0) / m0, same), flush=True)
