"""Host time to ENQUEUE one training step (ctypes launches, descriptor bookkeeping) against its GPU time, in the three feed modes
of bench.py (plain / in-batch de-duplication / + frozen-layer cache).  The fp16 loss scaler is off here: its poll waits for the
step two back, which would make the enqueue time read as GPU time.      GPU box: python tools/scratch/host_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch
import engine as E, hashinit, synth
from dedup import build_plan
from schema import FULL, state_shapes
dev, B, N_NEWS, seed = "cuda:0", 32, 51282, 1234
cfg = E.EngineConfig(n_layers=4, trainable_layers=(2, 3), num_teachers=4)
eng = E.Engine(cfg, dev, max_batch=B, dtype="fp16")
eng.load_state_dict(hashinit.init_state_dict(seed, state_shapes(FULL, 4, cfg.D, 4)))
eng.scaler.enabled = False
comb = torch.from_numpy(synth.news_table(seed, N_NEWS, cfg.L)).to(dev)
tables = torch.from_numpy(synth.teacher_tables(seed, 4, N_NEWS, cfg.D)).to(dev)
S = 40
hidx, mask, cidx, label = [torch.from_numpy(x).to(dev) for x in synth.impressions(seed + 1, S * B, N_NEWS, cfg.U, cfg.C)]
hn, cn = hidx.cpu().numpy(), cidx.cpu().numpy()
plans = [build_plan(hn[i * B:(i + 1) * B], cn[i * B:(i + 1) * B]).to(dev) for i in range(S)]
def step(i, plan):
    s = slice(i * B, (i + 1) * B)
    eng.forward_indexed(comb, hidx[s], mask[s], cidx[s], label[s], tables, plans[i] if plan else None)
    eng.backward()
    eng.step(lr=1e-4)
for mode in ("plain", "dedup", "dedup + frozen-layer cache"):
    plan = mode != "plain"
    if mode.endswith("cache"):
        eng.build_frozen_cache(comb)
    for i in range(5): step(i, plan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(S): step(i, plan)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-28s enqueue %.2f ms/step (host) ; wall %.2f ms/step ; the GPU queue is %s" % (
        mode, (t1 - t0) / S * 1e3, (t2 - t0) / S * 1e3, "never empty" if (t1 - t0) < 0.9 * (t2 - t0) else "at risk of running dry"))
