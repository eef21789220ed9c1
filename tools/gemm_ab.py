"""Interleaved same-process A/B of a library GEMM option (tnr_gemm_set_option) over the encoder's NT shapes; GPU box.
Box-to-box variance on this pool is up to 15 %, so only interleaved same-box comparisons are meaningful.
    AB=pp:0:1 python tools/gemm_ab.py        (two-phase main loop vs ping-pong)      DTYPE=fp16|bf16   M=rows   OPT=key=val,... (fixed for both sides)
    PROBE=8 ... : the same A/B in the probe build (tools/_probe, -DTNR_PROBES=2) with that probe set, e.g. 8 = K loops without epilogues"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
if os.environ.get("LIB"):                 # another build of the library (e.g. tools/_noepi: tools/probes/build.sh _noepi -DTNR_NOEPI)
    T.LIB_PATH = os.path.join(ROOT, os.environ["LIB"], "libtnr_hip.so")
if os.environ.get("PROBE"):
    T.LIB_PATH = os.path.join(ROOT, "tools", "_probe", "libtnr_hip.so")
    T.lib().tnr_gemm_set_option(b"probe", int(os.environ["PROBE"]))
dev = "cuda:0"
M = int(os.environ.get("M", 52800))
KEY, *VALS = os.environ.get("AB", "pp:0:1").split(":")
for kv in filter(None, os.environ.get("OPT", "").split(",")):          # options fixed for both sides, e.g. OPT=allow_fine=0
    T.lib().tnr_gemm_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1]))
F16 = os.environ.get("DTYPE", "fp16") == "fp16"
td, sfx = (torch.float16, "_f16") if F16 else (torch.bfloat16, "")
SHAPES = ((3072, 768, 0), (3072, 768, 67), (3072, 768, 3), (3072, 768, 16 | 128), (768, 3072, 9), (2304, 768, 1), (768, 768, 9), (768, 768, 0),
          (768, 2304, 8), (768, 3072, 8), (256, 768, 1 | 4 | 32), (768, 256, 8))
tot = [0.0, 0.0]
for (N, K, fl) in SHAPES:
    a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
    c = torch.zeros((M, N), device=dev, dtype=torch.float32 if fl & 32 else td)
    bias = torch.randn(N, device=dev); r = torch.randn((M, N), device=dev).to(td); aux = torch.randn((M, N), device=dev).to(td)
    cs = torch.zeros((T.query("tnr_gemm_colsum_rows", M), N), device=dev) if fl & 128 else None
    def run():
        T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, bias, r if fl & 8 else None, N if fl & 8 else 0,
               aux if fl & (64 | 16) else None, N if fl & (64 | 16) else 0, fl, cs)
    acc = collections.defaultdict(list)
    for rnd in range(8):
        for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            T.lib().tnr_gemm_set_option(KEY.encode(), int(VALS[v]))
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            acc[v].append(e0.elapsed_time(e1) * 100)
    m0, m1 = sorted(acc[0])[4], sorted(acc[1])[4]
    tot[0] += m0; tot[1] += m1
    print("N=%4d K=%4d flags %3d route %4d: %s=%s %.1f us (%.0f TF)   %s=%s %.1f us (%.0f TF)   (%+.1f %%)" % (
        N, K, fl, T.query("tnr_gemm_nt_route" + sfx, M, N, K, fl), KEY, VALS[0], m0, 2.0 * M * N * K / m0 / 1e6, KEY, VALS[1], m1,
        2.0 * M * N * K / m1 / 1e6, 100 * (m1 - m0) / m0), flush=True)
print("sum: %.1f us vs %.1f us (%+.1f %%)" % (tot[0], tot[1], 100 * (tot[1] - tot[0]) / tot[0]))
