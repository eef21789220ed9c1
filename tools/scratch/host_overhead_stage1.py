"""Host enqueue time against GPU time of one stage-1 step (bench.py's "configs[4] stage 1" leg: two encoder passes per step);
8 steps per measurement so that the launch queue never fills.   GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import numpy as np, torch
import hashinit, synth
from stage1 import Stage1Engine
dev, B, seed = "cuda:0", 32, 1234
Lt, Lb, Kn, nd = 30, 128, 4, 20000
s1 = Stage1Engine(n_layers=2, trainable_layers=(0, 1), num_teachers=4, npratio=Kn, title_len=Lt, body_len=Lb, device=dev, batch=B, dtype="fp16")
s1.load_state_dict({k: torch.from_numpy(hashinit.init_tensor(seed, k, tuple(sh))) for k, sh in s1.shapes.items()})
s1.title.refresh_shadows(all_layers=True)
s1.body.refresh_rel()
s1.title.scaler.enabled = False
d_title = torch.from_numpy(synth.news_table(11, nd - 1, Lt)).to(dev)
d_body = torch.from_numpy(synth.news_table(12, nd - 1, Lb, mean_len=0.6 * Lb, std_len=0.25 * Lb)).to(dev)
d_tt = torch.from_numpy(np.ascontiguousarray(synth.teacher_tables(13, 4, nd - 1, s1.cfg_t.D))).to(dev)
d_tb = torch.from_numpy(np.ascontiguousarray(synth.teacher_tables(14, 4, nd - 1, s1.cfg_t.D))).to(dev)
rs = np.random.RandomState(seed)
S = 8
pidx = torch.from_numpy(rs.randint(1, nd, ((S + 5) * B, 1 + Kn)).astype(np.int32)).to(dev)
lab1 = torch.zeros(B, dtype=torch.int64, device=dev)
import tnr_hip as T
n_calls = [0]
orig = T.call
def counting(*a, **k):
    n_calls[0] += 1
    return orig(*a, **k)
def st1(i):
    s1.forward_indexed(d_title, d_body, pidx[i * B:(i + 1) * B], lab1, d_tt, d_tb)
    s1.backward()
    s1.step(1e-5, lr_bert=1e-6, amsgrad=False)
for i in range(5): st1(i)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5, 5 + S): st1(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("stage 1: enqueue %.2f ms/step (host) ; wall %.2f ms/step" % ((t1 - t0) / S * 1e3, (t2 - t0) / S * 1e3))
