"""A few steps of the bench workload with the teacher side on a second stream, for `rocprofv3 --kernel-trace` (GPU box):
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ts_trace -- python3 tools/scratch/ts_trace.py [0|1]"""
import os, sys
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
S = 8
hidx, mask, cidx, label = [torch.from_numpy(x).to(dev) for x in synth.impressions(seed + 1, S * B, N_NEWS, cfg.U, cfg.C)]
if len(sys.argv) > 1 and int(sys.argv[1]):
    eng.teacher_stream = torch.cuda.Stream()
for i in range(S):
    s = slice(i * B, (i + 1) * B)
    eng.forward_indexed(comb, hidx[s], mask[s], cidx[s], label[s], tables)
    eng.backward()
    eng.step(lr=1e-4)
torch.cuda.synchronize()
