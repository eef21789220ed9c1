"""Interleaved same-process A/B of a library GEMM option on the WHOLE training step (the bench.py workload: B=32, 4-layer
student + 4 teachers, fwd + bwd + AMSGrad); GPU box.     AB=pp:0:1 DTYPE=fp16 python tools/step_ab.py
AB=teacher_stream:0:1: the engine's second stream for the teacher side instead of a library option."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import numpy as np, torch
import engine as E, hashinit, synth, tnr_hip as T
from schema import FULL, state_shapes
KEY, *VALS = os.environ.get("AB", "pp:0:1").split(":")
dtype = os.environ.get("DTYPE", "fp16")
dev, B, N_NEWS, seed = "cuda:0", 32, 51282, 1234
cfg = E.EngineConfig(n_layers=4, trainable_layers=(2, 3), num_teachers=4)
eng = E.Engine(cfg, dev, max_batch=B, dtype=dtype)
eng.load_state_dict(hashinit.init_state_dict(seed, state_shapes(FULL, 4, cfg.D, 4)))
comb = torch.from_numpy(synth.news_table(seed, N_NEWS, cfg.L)).to(dev)
tables = torch.from_numpy(synth.teacher_tables(seed, 4, N_NEWS, cfg.D)).to(dev)
S = 20
hidx, mask, cidx, label = [torch.from_numpy(x).to(dev) for x in synth.impressions(seed + 1, S * B, N_NEWS, cfg.U, cfg.C)]
def step(i):
    s = slice(i * B, (i + 1) * B)
    eng.forward_indexed(comb, hidx[s], mask[s], cidx[s], label[s], tables)
    eng.backward()
    eng.step(lr=1e-4)
side = torch.cuda.Stream()
res = {0: [], 1: []}
for rnd in range(6):
    for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
        if KEY == "teacher_stream":        # engine-level switch: the teacher side of the forward on a second stream
            eng.teacher_stream = side if int(VALS[v]) else None
        elif KEY == "sg_kdiv":             # engine-level: K per split of the small fp32 GEMMs
            eng.sg_kdiv = int(VALS[v])
        elif KEY == "merge_reductions":    # engine-level: one batched partial-sum reduction per backward (1) or one per gradient bucket (0)
            eng.merge_reductions = bool(int(VALS[v]))
        elif KEY == "group_wgrad":         # engine-level: a layer's four weight gradients in one persistent launch (1) or one launch each (0)
            eng.group_wgrad = bool(int(VALS[v]))
        elif KEY == "group_sgemm":         # engine-level: independent fp32 GEMMs of the heads in one launch (1) or one launch each (0)
            eng.group_sgemm = bool(int(VALS[v]))
        else:
            T.lib().tnr_gemm_set_option(KEY.encode(), int(VALS[v]))
        for i in range(3): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(S): step(i)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / S * 1e3)
m0, m1 = sorted(res[0])[len(res[0]) // 2], sorted(res[1])[len(res[1]) // 2]
print("%s step: %s=%s %.3f ms (%.0f imp/s)   %s=%s %.3f ms (%.0f imp/s)   (%+.1f %%)" % (dtype, KEY, VALS[0], m0, B / m0 * 1e3, KEY, VALS[1], m1, B / m1 * 1e3, 100 * (m1 - m0) / m0))
print("all:", [round(x, 3) for x in res[0]], [round(x, 3) for x in res[1]])
