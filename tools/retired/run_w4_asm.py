"""Check and time the hand-scheduled four-wave NT GEMM tile kernel (tools/w4_proto/gen_w4_hip.py -> libw4asm.so) against the shipped
eight-wave kernel (tools only).
    python tools/w4_proto/gen_w4_hip.py [knobs] > /tmp/w4.hip && hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/w4_proto/libw4asm.so /tmp/w4.hip
    python tools/w4_proto/run_w4_asm.py [N K]           (LIB=<path> for another build of the prototype; SYM=w4r_launch for
    gen_w4r_hip.py's register-staged form; PROBE_LIB=tools/_noepi/libtnr_hip.so adds the shipped kernel without its epilogue)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.environ.get("LIB", os.path.join(HERE, "libw4asm.so")))
SYM = os.environ.get("SYM", "w4_launch")             # gen_w4r_hip.py: SYM=w4r_launch
getattr(L, SYM).argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
dev = "cuda:0"
VAR = {"full": 0, "mfma": 1, "noread": 2, "nodma": 3, "noepi": 4 if SYM == "w4r_launch" else 1}
VARIANTS = os.environ.get("VARIANTS", "full,mfma,nodma,noread").split(",")


def launch(v, a, b, c, M, N, K):
    rc = getattr(L, SYM)(VAR[v], a.data_ptr(), b.data_ptr(), c.data_ptr(), a.stride(0) * 2, b.stride(0) * 2, c.stride(0) * 2, M, N, K,
                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def check(M, N, K):
    torch.manual_seed(0)
    a = (torch.randn((M, K), device=dev) * 0.5).to(torch.float16)
    b = (torch.randn((N, K), device=dev) * 0.05).to(torch.float16)
    c = torch.full((M, N), 7.0, device=dev, dtype=torch.float16)
    guard = torch.full((1 << 20,), 3.0, device=dev, dtype=torch.float16)        # allocated right behind C: a stray store shows up here
    launch("full", a, b, c, M, N, K)
    torch.cuda.synchronize()
    c2 = torch.zeros_like(c)
    T.call("tnr_gemm_nt_ex_f16", a, K, b, K, c2, N, M, N, K, None, None, 0, None, 0, 0, None)
    torch.cuda.synchronize()
    ref = a.float() @ b.float().t() if M <= 4096 else None
    err = float((c.float() - ref).abs().max()) if ref is not None else float("nan")
    same = bool(torch.equal(c, c2))
    print("M=%d N=%d K=%d: max |err| vs fp32 %.3e ; bit-identical to the shipped kernel: %s ; guard intact: %s" % (
        M, N, K, err, same, bool((guard == 3.0).all())), flush=True)
    if not same:
        bad = (c != c2)
        print("   differing: %d of %d ; rows %s ; cols %s" % (int(bad.sum()), c.numel(), bad.any(1).nonzero().flatten()[:12].tolist(),
                                                              bad.any(0).nonzero().flatten()[:12].tolist()), flush=True)
    return same, a, b, c, c2


def main():
    N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (768, 3072)
    ok, *_ = check(512, N, 256)                      # small first: nothing large runs on a kernel that is wrong at 6 tiles
    if not ok:
        sys.exit(1)
    ok, *_ = check(700, N, K)                        # rows that are not a multiple of the tile height
    if not ok:
        sys.exit(1)
    M = int(os.environ.get("M", 52800))
    ok, a, b, c, c2 = check(M, N, K)
    if not ok:
        sys.exit(1)
    tiles = ((M + 255) // 256) * (N // 256)
    rounds = -(-tiles // 256)
    res = {}
    for rep in range(3):                                      # interleaved: prototype variants and the shipped kernel in turn
        for v in VARIANTS:
            res.setdefault(v, []).append(timeit(lambda: launch(v, a, b, c, M, N, K)))
        res.setdefault("shipped 8-wave (plain store)", []).append(
            timeit(lambda: T.call("tnr_gemm_nt_ex_f16", a, K, b, K, c2, N, M, N, K, None, None, 0, None, 0, 0, None)))
        if os.environ.get("PROBE_LIB"):                       # the shipped kernel without its epilogue (probe 8 of a -DTNR_PROBES=2 build)
            if rep == 0:
                import importlib.util
                spec = importlib.util.spec_from_file_location("tnr_probe", os.path.join(ROOT, "tiny-newsrec_amd", "tnr_hip.py"))
                TP = importlib.util.module_from_spec(spec); spec.loader.exec_module(TP)
                TP.LIB_PATH = os.path.join(ROOT, os.environ["PROBE_LIB"])
                if "_noepi" not in TP.LIB_PATH:               # tools/_noepi (-DTNR_NOEPI) has no epilogue at all: the clean K loop
                    TP.lib().tnr_gemm_set_option(b"probe", 8)
            res.setdefault("shipped 8-wave, no epilogue", []).append(
                timeit(lambda: TP.call("tnr_gemm_nt_ex_f16", a, K, b, K, c2, N, M, N, K, None, None, 0, None, 0, 0, None)))
    res = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    for k, us in res.items():
        print("   %-32s %8.1f us   %6.0f TF   per K step (%.2f rounds of tiles) %.2f us" % (k, us, 2.0 * M * N * K / us / 1e6, tiles / 256.0, us / rounds / (K // 64)), flush=True)


if __name__ == "__main__":
    main()
