"""Per-step errors of the engine's replay of tests/golden/trajectory_0.npz (the reference's 50 steps); GPU box.
DTYPE=fp16|bf16 python tools/scratch/traj_detail.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "tiny-newsrec_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import engine as E
from helpers import load_trajectory_case
import test_engine_gpu as TG
dtype = os.environ.get("DTYPE", "fp16")
z, P, cfg, batches, lr, steps = load_trajectory_case()
eng, B = TG._engine_for(cfg, z, len(batches[0][4]), dtype)
eng.load_state_dict(P)
dev = [TG._dev_inputs(b) for b in batches]
for step in range(steps):
    losses, score = eng.forward(*dev[step % len(dev)])
    eng.backward(); eng.step(lr)
    l = losses.cpu().numpy()
    got = np.array([l[0] + cfg["coef"] * l[1] + l[2], l[0], l[2], l[1]])
    sc = score.cpu().numpy()
    print("%2d b%d  loss err tot %.1e dis %.1e emb %.1e tgt %.1e | ref tot %.3f tgt %.3f | score err %.1e  |score| %.2f" % (
        step, step % len(dev), *np.abs(got - z["losses"][step]), z["losses"][step][0], z["losses"][step][3],
        np.abs(sc - z["scores"][step]).max(), np.abs(z["scores"][step]).max()))
