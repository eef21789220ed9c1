"""Per-shape timing of the GEMM entry points (development aid; run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch
import tnr_hip as T

dev = "cuda:0"
M = int(os.environ.get("M", 52800))
reps = 10


def bench_nt(N, K, flags, name):
    pa, pb = int(os.environ.get("PAD_A", 0)), int(os.environ.get("PAD_B", 0))     # leading-dimension padding (elements)
    a = (torch.randn((M, K + pa), device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn((N, K + pb), device=dev) * 0.05).to(torch.bfloat16)
    c = torch.zeros((M, N), device=dev, dtype=torch.float32 if flags & T.EPI_OUTF32 else torch.bfloat16)
    bias = torch.randn(N, device=dev)
    res = torch.randn((M, N), device=dev).to(torch.bfloat16) if flags & T.EPI_RES else None
    aux = torch.randn((M, N), device=dev).to(torch.bfloat16) if flags & (T.EPI_AUXOUT | T.EPI_MULDGELU) else None
    def run():
        T.call("tnr_gemm_nt", a, K + pa, b, K + pb, c, N, M, N, K, bias, res, N if res is not None else 0, aux, N if aux is not None else 0, flags)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("NT %-22s N=%5d K=%5d  %8.1f us  %7.1f TF" % (name, N, K, us, 2.0 * M * N * K / us / 1e6))


def bench_tn(N, K, splits):
    Mp = (M + 63) // 64 * 64
    dy = torch.zeros((Mp, N), device=dev, dtype=torch.bfloat16); dy[:M] = torch.randn((M, N), device=dev).to(torch.bfloat16)
    x = torch.zeros((Mp, K), device=dev, dtype=torch.bfloat16); x[:M] = torch.randn((M, K), device=dev).to(torch.bfloat16)
    ws = torch.zeros(T.query("tnr_gemm_tn_ws_elems", N, K, splits), device=dev)
    dw = torch.zeros((N, K), device=dev)
    def run():
        T.call("tnr_gemm_tn_wgrad", dy, N, x, K, dw, K, M, N, K, ws, splits, 0)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("TN N=%5d K=%5d splits=%2d  %8.1f us  %7.1f TF" % (N, K, splits, us, 2.0 * M * N * K / us / 1e6))


if __name__ == "__main__":
    print("M =", M, "ver", os.environ.get("TNR_GEMM_VER", "3"))
    if os.environ.get("ONLY"):
        reps = 3
        for item in os.environ["ONLY"].split(","):
            kind, N, K, fl = item.split(":")
            if kind == "nt":
                bench_nt(int(N), int(K), int(fl), "only")
            else:
                bench_tn(int(N), int(K), int(fl))
        sys.exit(0)
    bench_nt(2304, 768, T.EPI_BIAS, "qkv bias")
    bench_nt(768, 768, T.EPI_BIAS | T.EPI_RES, "attn-out bias+res")
    bench_nt(3072, 768, T.EPI_BIAS | T.EPI_GELU, "ffn-up gelu")
    bench_nt(3072, 768, T.EPI_BIAS | T.EPI_GELU | T.EPI_AUXOUT, "ffn-up gelu+aux")
    bench_nt(3072, 768, 0, "ffn-up plain")
    bench_nt(768, 3072, T.EPI_BIAS | T.EPI_RES, "ffn-down bias+res")
    bench_nt(768, 3072, 0, "ffn-down plain")
    bench_nt(3072, 768, T.EPI_MULDGELU, "dgrad w2 *dgelu")
    bench_nt(768, 2304, T.EPI_RES, "dgrad qkv +res")
    bench_nt(256, 768, T.EPI_BIAS | T.EPI_TANH | T.EPI_OUTF32, "pool fc1 tanh f32")
    bench_nt(768, 256, T.EPI_RES, "pool dgrad")
    for N, K in ((3072, 768), (768, 3072), (2304, 768), (768, 768), (256, 768)):
        for sp in (4, 8, 16):
            bench_tn(N, K, sp)
