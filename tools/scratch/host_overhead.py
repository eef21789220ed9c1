"""Host time to ENQUEUE one training step (ctypes launches, descriptor bookkeeping) against its GPU time; GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch
import engine as E, hashinit, synth
from schema import FULL, state_shapes
dev, B, N_NEWS, seed = "cuda:0", 32, 51282, 1234
cfg = E.EngineConfig(n_layers=4, trainable_layers=(2, 3), num_teachers=4)
eng = E.Engine(cfg, dev, max_batch=B, dtype="fp16")
eng.load_state_dict(hashinit.init_state_dict(seed, state_shapes(FULL, 4, cfg.D, 4)))
comb = torch.from_numpy(synth.news_table(seed, N_NEWS, cfg.L)).to(dev)
tables = torch.from_numpy(synth.teacher_tables(seed, 4, N_NEWS, cfg.D)).to(dev)
S = 30
hidx, mask, cidx, label = [torch.from_numpy(x).to(dev) for x in synth.impressions(seed + 1, S * B, N_NEWS, cfg.U, cfg.C)]
def step(i):
    s = slice(i * B, (i + 1) * B)
    eng.forward_indexed(comb, hidx[s], mask[s], cidx[s], label[s], tables)
    eng.backward()
    eng.step(lr=1e-4)
for i in range(5): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(S): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/step (host) ; wall %.2f ms/step ; the GPU queue is %s" % ((t1 - t0) / S * 1e3, (t2 - t0) / S * 1e3,
      "never empty" if (t1 - t0) < 0.9 * (t2 - t0) else "at risk of running dry"))
