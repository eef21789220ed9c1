"""Repeat the 12-layer PLM-NR two-step golden under library options; print score errors (determinism / race screen)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tiny-newsrec_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import engine as E, tnr_hip as T
from helpers import load_plmnr_case
z, P, cfg, inp = load_plmnr_case("plmnr_full_1.npz")
seed, B, _, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
hist, mask, cand, label = [t(x) for x in inp]
lr_bert, lr = [float(x) for x in z["lrs"]]
for opts in ({"tnpp": 0, "pp": 0}, {"tnpp": 1, "pp": 0}, {"tnpp": 0, "pp": 1}, {"tnpp": 1, "pp": 1}):
    for k, v in opts.items():
        T.lib().tnr_gemm_set_option(k.encode(), v)
    for rep in range(3):
        ec = E.EngineConfig(n_layers=nl, trainable_layers=cfg["trainable_layers"], num_teachers=0, user_log_length=U, npratio=C - 1,
                            num_words=L, news_dim=D, user_log_mask=False, temperature=1.0, coef=1.0)
        eng = E.Engine(ec, "cuda:0", max_batch=B, dtype="fp16")
        eng.load_state_dict(P)
        errs, gsum = [], None
        for step in range(2):
            losses, score = eng.forward(hist, mask, cand, label)
            errs.append(float(np.abs(score.cpu().numpy() - z["score%d" % step]).max()))
            eng.backward()
            if step == 0:
                gsum = float(eng.flat_g.double().abs().sum())
            eng.step(lr, lr_bert=lr_bert)
        print(opts, "rep", rep, "score err step0 %.4e step1 %.4e  |g|_1 %.10e" % (errs[0], errs[1], gsum), flush=True)
