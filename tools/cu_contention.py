"""What happens to the step's GEMMs while another stream holds some CUs (a collective kernel during the overlapped gradient
all-reduce)?  tnr_debug_cu_hog(n_wg, us) (libtnr_testhooks.so, a test-only library built beside the product one): n_wg workgroups that each own a whole CU and spin.
Times NT / TN launches alone and beside 8 / 16 / 32 held CUs.  LIB=tools/_probe/libtnr_old.so: the same for another build."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
L = ctypes.CDLL(os.path.join(os.path.dirname(T.LIB_PATH), "libtnr_testhooks.so"))   # the hog comes from the test-hooks library ...
L.tnr_debug_cu_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
if os.environ.get("LIB"): T.LIB_PATH = os.path.join(ROOT, os.environ["LIB"])  # ... the GEMMs from the shipped one, or from LIB=<another build>
T.lib()
dev, M, td, sfx = "cuda:0", 52800, torch.float16, "_f16"
side = torch.cuda.Stream()
import engine as E
MULT = int(os.environ.get("UNITS", 1))          # weight gradient: UNITS x the single-GPU split count (finer work units for the queue)
for (kind, N, K) in (("nt", 3072, 768), ("nt", 768, 3072), ("tn", 3072, 768), ("tn", 768, 768), ("tn", 2304, 768)):
    if kind == "nt":
        a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
        c = torch.zeros((M, N), device=dev, dtype=td)
        run = lambda: T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, None, None, 0, None, 0, 0, None)
    else:
        Mp = (M + 127) // 128 * 128
        dy = torch.zeros((Mp, N), device=dev, dtype=td); x = torch.zeros((Mp, K), device=dev, dtype=td)
        dy[:M] = (torch.randn((M, N), device=dev) * 0.1).to(td); x[:M] = torch.randn((M, K), device=dev).to(td)
        dw = torch.zeros((N, K), device=dev); sp = min(64, E.Engine._wgrad_splits(N, K)[0] * MULT); ws = torch.zeros(N * K * sp, device=dev)
        run = lambda: T.call("tnr_gemm_tn_wgrad" + sfx, dy, N, x, K, dw, K, M, N, K, ws, sp, 0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    res = []
    for hog in (0, 8, 16, 32):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize()
            if hog:
                L.tnr_debug_cu_hog(hog, 3000, side.cuda_stream)          # holds `hog` CUs for 3 ms on the side stream
            torch.cuda._sleep(200000)                                    # let it start first (and, alone too: clocks up after the idle sync)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 250)
        res.append(sorted(ts)[2])
    print("%s N=%d K=%d: alone %.1f us ; beside 8 / 16 / 32 held CUs %.1f / %.1f / %.1f us (x%.2f / x%.2f / x%.2f)" % (
        kind, N, K, res[0], res[1], res[2], res[3], res[1] / res[0], res[2] / res[0], res[3] / res[0]), flush=True)
