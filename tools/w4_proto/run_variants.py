import ctypes, os, sys, torch
L = ctypes.CDLL(sys.argv[1]); f = L.w4r_launch
f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
M, N, K = 52800, int(sys.argv[2]), int(sys.argv[3])
a = (torch.randn((M, K), device="cuda") * 0.5).half(); b = (torch.randn((N, K), device="cuda") * 0.05).half(); c = torch.zeros((M, N), device="cuda", dtype=torch.half)
def t(v, n=20):
    for _ in range(3): assert f(v, a.data_ptr(), b.data_ptr(), c.data_ptr(), K * 2, K * 2, N * 2, M, N, K, torch.cuda.current_stream().cuda_stream) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f(v, a.data_ptr(), b.data_ptr(), c.data_ptr(), K * 2, K * 2, N * 2, M, N, K, torch.cuda.current_stream().cuda_stream)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
for name, v in (("full", 0), ("noepi", 4), ("mfma", 1), ("noread", 2), ("nodma", 3)):
    print("%-8s %.1f us" % (name, sorted(t(v) for _ in range(3))[1]))
