"""How much of each NT launch of the headline step is its epilogue?  Probe build (tools/_probe, -DTNR_PROBES=2): every (N, K,
flags) of the step timed in full, without the epilogue (probe 8) and with the epilogue's arithmetic but none of its stores
(probe 16) -- the upper bound of what overlapping the epilogue with the K loop can give.  Interleaved, median of 5 x 10 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
if not os.environ.get("PRODUCT"):
    T.LIB_PATH = os.path.join(ROOT, "tools", "_probe", "libtnr_hip.so")
dev, M = "cuda:0", int(os.environ.get("M", 52800))
td, sfx = torch.float16, "_f16"
opt = lambda k, v: T.lib().tnr_gemm_set_option(k.encode(), int(v))
H, I, Q = 768, 3072, 256
B_, G_, TH, R_, MD, F32O, AUX, CS = (T.EPI_BIAS, T.EPI_GELU, T.EPI_TANH, T.EPI_RES, T.EPI_MULDGELU, T.EPI_OUTF32, T.EPI_AUXOUT, T.EPI_COLSUM)
LAUNCHES = [("qkv", 3 * H, H, B_, 4), ("attn_out", H, H, B_ | R_, 4), ("ffn_up_kept", I, H, B_ | G_ | AUX, 2), ("ffn_up_frozen", I, H, B_ | G_, 2),
            ("ffn_down", H, I, B_ | R_, 4), ("pool_fc1", Q, H, B_ | TH | F32O, 1), ("dgrad_pool", H, Q, R_, 1),
            ("dgrad_w2_gelu'", I, H, MD | CS, 2), ("dgrad_w1", H, I, R_, 2), ("dgrad_o", H, H, 0, 2), ("dgrad_qkv", H, 3 * H, R_, 1)]
PROBES = [(0, "full")] + ([] if os.environ.get("PRODUCT") else [(8, "no epilogue"), (16, "no stores"), (8 | 128, "no epilogue + 16 stores per wave spread over the K loop")])
if os.environ.get("SPLIT"):               # what the arithmetic half of the epilogue is made of: table lookups / operand loads off
    PROBES = [(8, "no epilogue"), (16, "no stores"), (16 | 2048, "no stores, no table"), (16 | 4096, "no stores, no operand loads"), (16 | 2048 | 4096, "neither")]
if os.environ.get("AB"):                  # AB=pp:1:2 -> interleaved A/B of a library option instead of the probes (PRODUCT=1)
    key, va, vb = os.environ["AB"].split(":")
    PROBES = [(int(va), "%s=%s" % (key, va)), (int(vb), "%s=%s" % (key, vb))]
PKEY = os.environ["AB"].split(":")[0] if os.environ.get("AB") else "probe"
if os.environ.get("CUS"):                 # CUS=128 M=26400: the same tiles per CU on half the chip - does a launch-wide burst of epilogue traffic cost?
    opt("cus", int(os.environ["CUS"]))
tot = {n: 0.0 for _, n in PROBES}
for name, N, K, fl, cnt in LAUNCHES:
    a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
    c = torch.zeros((M, N), device=dev, dtype=torch.float32 if fl & F32O else td)
    bias = torch.randn(N, device=dev); r = torch.randn((M, N), device=dev).to(td)
    aux = torch.randn((M, N), device=dev).to(td) if fl & (MD | AUX) else None
    cs = torch.zeros((T.query("tnr_gemm_colsum_rows" + sfx, M), N), device=dev) if fl & CS else None
    def run():
        T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, bias if fl & B_ else None, r if fl & R_ else None, N if fl & R_ else 0,
               aux, N if aux is not None else 0, fl, cs)
    res = {}
    for rep in range(5):
        for p, pn in PROBES:
            opt(PKEY, p)
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(pn, []).append(e0.elapsed_time(e1) * 100)
    opt(PKEY, PROBES[-1][0] if os.environ.get("AB") else 0)
    line = "%-16s N=%4d K=%4d x%d:" % (name, N, K, cnt)
    for p, pn in PROBES:
        us = sorted(res[pn])[2]
        tot[pn] += us * cnt
        line += "  %s %6.1f us (%4.0f TF)" % (pn, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
print("step total (25 launches):", "  ".join("%s %.0f us" % (n, v) for n, v in tot.items()))
