"""What the fp16 loss scaler costs per step: the bench workload with the scaler on / off (fp16) and the bf16 build, interleaved
blocks of 50 steps, medians.   python tools/scratch/scaler_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch
import engine as E, hashinit, synth
from schema import FULL, state_shapes
dev, B, N_NEWS, seed = "cuda:0", 32, 51282, 1234
cfg = E.EngineConfig(n_layers=4, trainable_layers=(2, 3), num_teachers=4)
comb = torch.from_numpy(synth.news_table(seed, N_NEWS, cfg.L)).to(dev)
tables = torch.from_numpy(synth.teacher_tables(seed, 4, N_NEWS, cfg.D)).to(dev)
hidx, mask, cidx, label = [torch.from_numpy(x).to(dev) for x in synth.impressions(seed + 1, 40 * B, N_NEWS, cfg.U, cfg.C)]
engs = {}
for name, dt in (("fp16 scaler on", "fp16"), ("fp16 scaler off", "fp16"), ("bf16", "bf16")):
    e = E.Engine(cfg, dev, max_batch=B, dtype=dt)
    e.load_state_dict(hashinit.init_state_dict(seed, state_shapes(FULL, 4, cfg.D, 4)))
    if name.endswith("off"):
        e.scaler.enabled = False
    engs[name] = e
def block(eng, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        s = slice((i % 40) * B, (i % 40 + 1) * B)
        eng.forward_indexed(comb, hidx[s], mask[s], cidx[s], label[s], tables)
        eng.backward()
        eng.step(lr=1e-4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
res = {k: [] for k in engs}
for k, e in engs.items(): block(e, 10)
for rnd in range(7):
    for k, e in (list(engs.items()) if rnd % 2 == 0 else list(engs.items())[::-1]):
        res[k].append(block(e, 50))
for k, v in res.items():
    print("%-18s median %.3f ms/step  (%s)" % (k, sorted(v)[len(v) // 2], " ".join("%.3f" % x for x in v)))
