"""Assemble (tools/w4_proto/gen_w4_asm.py) and time the hand-scheduled four-wave NT GEMM tile kernel against the shipped eight-wave
kernel (tools only).  python tools/w4_proto/run_w4_asm.py [N K] [knobs]     e.g.  768 3072 dma_step=4,bar_at=32"""
import ctypes, os, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
HERE = os.path.dirname(os.path.abspath(__file__))
LLVM = "/opt/rocm/lib/llvm/bin"
hip = ctypes.CDLL("libamdhip64.so")
dev = "cuda:0"


def build(variant, knobs=""):
    tag = variant + ("_" + knobs.replace("=", "").replace(",", "_") if knobs else "")
    s, o, h = ["/tmp/w4_%s.%s" % (tag, e) for e in ("s", "o", "hsaco")]
    src = subprocess.run([sys.executable, os.path.join(HERE, "gen_w4_asm.py"), variant] + ([knobs] if knobs else []), capture_output=True, text=True, check=True).stdout
    open(s, "w").write(src)
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o], check=True)
    subprocess.run([LLVM + "/ld.lld", "-shared", o, "-o", h], check=True)
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipModuleLoad(ctypes.byref(mod), h.encode()) == 0
    assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, ("w4_gemm_" + variant).encode()) == 0
    return fn


def launch(fn, a, b, c, M, N, K):
    args = struct.pack("<QQQiiiiii", a.data_ptr(), b.data_ptr(), c.data_ptr(), a.stride(0) * 2, b.stride(0) * 2, c.stride(0) * 2, M, K // 64, N // 256)
    buf = ctypes.create_string_buffer(args, len(args))
    size = ctypes.c_size_t(len(args))
    extra = (ctypes.c_void_p * 5)(1, ctypes.cast(buf, ctypes.c_void_p), 2, ctypes.cast(ctypes.byref(size), ctypes.c_void_p), 3)
    grid = ((M + 255) // 256) * (N // 256)
    rc = hip.hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None, extra)
    assert rc == 0, rc


def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (768, 3072)
    knobs = sys.argv[3] if len(sys.argv) > 3 else ""
    M = int(os.environ.get("M", 52800))
    torch.manual_seed(0)
    a = (torch.randn((M, K), device=dev) * 0.5).to(torch.float16)
    b = (torch.randn((N, K), device=dev) * 0.05).to(torch.float16)
    c = torch.zeros((M, N), device=dev, dtype=torch.float16)
    full = build("full", knobs)
    launch(full, a, b, c, M, N, K)
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (400,))]).to(dev)
    ref = (a[rows].float() @ b.float().t())
    err = (c[rows].float() - ref).abs().max().item()
    c2 = torch.zeros_like(c)
    T.call("tnr_gemm_nt_ex_f16", a, K, b, K, c2, N, M, N, K, None, None, 0, None, 0, 0, None)
    torch.cuda.synchronize()
    same = torch.equal(c, c2)
    print("N=%d K=%d M=%d: max |err| vs fp32 on %d rows %.3e ; bit-identical to the shipped kernel: %s" % (N, K, M, rows.numel(), err, same), flush=True)
    tiles = ((M + 255) // 256) * (N // 256)
    rounds = -(-tiles // 256)
    res = {}
    for v in ("full", "mfma", "nodma", "noread"):
        fn = full if v == "full" else build(v, knobs)
        res[v] = timeit(lambda: launch(fn, a, b, c, M, N, K))
    res["shipped 8-wave (plain store)"] = timeit(lambda: T.call("tnr_gemm_nt_ex_f16", a, K, b, K, c2, N, M, N, K, None, None, 0, None, 0, 0, None))
    for k, us in res.items():
        print("   %-32s %8.1f us   %6.0f TF   per K step (%.2f rounds of tiles) %.2f us" % (k, us, 2.0 * M * N * K / us / 1e6, tiles / 256.0, us / rounds / (K // 64)), flush=True)


if __name__ == "__main__":
    main()
