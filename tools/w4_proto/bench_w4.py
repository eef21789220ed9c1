"""Four-wave prototype (libw4.so, see gemm_w4.hip) against the shipped NT kernel, plain store, fp16; GPU box.
   Build first: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DTNR_BUILD_F16 -o tools/w4_proto/libw4.so tools/w4_proto/gemm_w4.hip"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
W = ctypes.CDLL(os.path.join(ROOT, "tools", "w4_proto", "libw4.so"))
P, L, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
W.w4_gemm.argtypes = [P, L, P, L, P, L, L, L, L, I, P]
dev, M, td = "cuda:0", 52736, torch.float16
def t(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (N, K) in ((3072, 768), (2304, 768), (768, 768), (768, 2304), (768, 3072)):
    a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
    c0 = torch.zeros((M, N), device=dev, dtype=td); c1 = torch.zeros_like(c0); c2 = torch.zeros_like(c0)
    st = torch.cuda.current_stream().cuda_stream
    pp = t(lambda: T.call("tnr_gemm_nt_ex_f16", a, K, b, K, c0, N, M, N, K, None, None, 0, None, 0, 0, None))
    w0 = t(lambda: W.w4_gemm(a.data_ptr(), K, b.data_ptr(), K, c1.data_ptr(), N, M, N, K, 0, st))
    w1 = t(lambda: W.w4_gemm(a.data_ptr(), K, b.data_ptr(), K, c2.data_ptr(), N, M, N, K, 1, st))
    c3 = torch.zeros_like(c0)
    w3 = t(lambda: W.w4_gemm(a.data_ptr(), K, b.data_ptr(), K, c3.data_ptr(), N, M, N, K, 3, st))
    w2 = t(lambda: W.w4_gemm(a.data_ptr(), K, b.data_ptr(), K, c2.data_ptr(), N, M, N, K, 2, st))
    fl = 2.0 * M * N * K / 1e6
    print("N=%4d K=%4d: shipped 8-wave %.1f us (%.0f TF) | 4-wave pipelined %.1f us (%.0f TF) | 4-wave round-1 form %.1f us (%.0f TF) | MFMA-only probe %.1f us (%.0f TF) | DMA spread over the MFMA rows %.1f us (%.0f TF) | equal: %s %s" % (
        N, K, pp, fl / pp, w0, fl / w0, w1, fl / w1, w2, fl / w2, w3, fl / w3, bool(torch.equal(c0, c1)), bool(torch.equal(c0, c3))), flush=True)
