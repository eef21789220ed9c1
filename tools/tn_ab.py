"""Interleaved same-process A/B of a library option on the weight-gradient GEMMs of the headline step (engine's split counts);
GPU box.     AB=tnpp:0:1 DTYPE=fp16 python tools/tn_ab.py
LIB=tools/_tnp2 ... : another build of the library (a retired probe build: tools/retired/gemm_tn_probes_and_schedules.diff.txt: the kernel without its
fragment reads; 1 = without LDS-DMA after the first two m steps, 4 = without MFMAs, sums combine)"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
import engine as E
if os.environ.get("LIB"):
    T.LIB_PATH = os.path.join(ROOT, os.environ["LIB"], "libtnr_hip.so")
dev, M = "cuda:0", int(os.environ.get("M", 52800))
Mp = (M + 127) // 128 * 128
KEY, *VALS = os.environ.get("AB", "tnpp:0:1").split(":")
F16 = os.environ.get("DTYPE", "fp16") == "fp16"
td, sfx = (torch.float16, "_f16") if F16 else (torch.bfloat16, "")
tot = [0.0, 0.0]
for (N, K) in ((3072, 768), (768, 3072), (2304, 768), (768, 768), (256, 768)):
    dy = torch.zeros((Mp, N), device=dev, dtype=td); dy[:M] = (torch.randn((M, N), device=dev) * 0.1).to(td)
    x = torch.zeros((Mp, K), device=dev, dtype=td); x[:M] = torch.randn((M, K), device=dev).to(td)
    dw = torch.zeros((N, K), device=dev)
    sp = E.Engine._wgrad_splits(N, K)[0]
    ws = torch.zeros(T.query("tnr_gemm_tn_ws_elems" + sfx, N, K, sp), device=dev)
    run = lambda: T.call("tnr_gemm_tn_wgrad" + sfx, dy, N, x, K, dw, K, M, N, K, ws, sp, 0)
    acc = collections.defaultdict(list)
    for rnd in range(8):
        for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            T.lib().tnr_gemm_set_option(KEY.encode(), int(VALS[v]))
            run(); run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            acc[v].append(e0.elapsed_time(e1) * 100)
    outs = []
    for v in (0, 1):                                  # the two variants' results on the same random operands, bit for bit
        T.lib().tnr_gemm_set_option(KEY.encode(), int(VALS[v]))
        dw.zero_(); run(); torch.cuda.synchronize(); outs.append(dw.clone())
    same = torch.equal(outs[0], outs[1])
    m0, m1 = sorted(acc[0])[4], sorted(acc[1])[4]
    tot[0] += m0; tot[1] += m1
    print("dW %4d x %4d splits %2d: %s=%s %.1f us (%.0f TF)   %s=%s %.1f us (%.0f TF)   (%+.1f %%)  [incl. slab reduce]  bit-identical: %s" % (
        N, K, sp, KEY, VALS[0], m0, 2.0 * M * N * K / m0 / 1e6, KEY, VALS[1], m1, 2.0 * M * N * K / m1 / 1e6, 100 * (m1 - m0) / m0, same), flush=True)
print("sum: %.1f us vs %.1f us (%+.1f %%)" % (tot[0], tot[1], 100 * (tot[1] - tot[0]) / tot[0]))
