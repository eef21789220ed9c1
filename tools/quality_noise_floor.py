"""What two FP32 implementations of the same training run differ by at the end: oracle/torch_port.py (the fp32 torch-CPU port of the
reference step - same formulas, another order of summation than the reference's modules) trained on the quality golden's batches
from the same init, evaluated by the fp32 oracle the way run.py:219-379 does, against the reference's own numbers in
tests/golden/quality_0.npz.  This is the yardstick for tests/test_quality_gpu.py (ii): a 400-step run at B = 8 under Adam is
chaotic (elements whose gradient sits at the rounding floor take +-lr steps of random sign), so even fp32 against fp32 does not
reproduce the metrics to 0.1 pt - only evaluation on the SAME weights does (test (i)).   CPU only, ~15 minutes on 8 cores.
    python tools/quality_noise_floor.py [threads=8]  ->  profiles/r06_quality_noise_floor.json"""
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tiny-newsrec_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import hashinit                                  # noqa: E402
import metrics as PM                             # noqa: E402
from oracle import data_oracle as DO             # noqa: E402
from oracle import newsrec_oracle as O           # noqa: E402
from oracle import torch_port as TP              # noqa: E402
from schema import FULL, state_shapes            # noqa: E402


def main():
    torch.set_num_threads(int(dict(a.split("=") for a in sys.argv[1:]).get("threads", 8)))
    z = np.load(os.path.join(ROOT, "tests", "golden", "quality_0.npz"))
    seed, B, T_, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
    steps, lr = int(z["steps"][0]), float(z["lr"][0])
    P0 = hashinit.init_state_dict(seed, state_shapes(FULL, nl, D, T_))
    cfg = dict(n_layers=nl, heads=A, trainable_layers=[int(x) for x in z["trainable"]], user_log_mask=True, temperature=1.0, coef=0.2)
    comb = z["news_combined"].astype(np.int64)
    tabs = [z["table%d" % i] for i in range(T_)]
    news_index = {"N%d" % i: i for i in range(1, comb.shape[0])}
    trn = TP.Trainer(P0, cfg, lr=lr)
    lines = [str(l).encode() for l in z["train_lines"]]
    losses, t0 = np.zeros(steps), time.time()
    for step in range(steps):
        h, m, c, y = DO.decode_batch(lines[step * B:(step + 1) * B], news_index, U, C - 1, labels=z["labels"][step])
        out = trn.step(comb[h], m, comb[c], y, [t[h] for t in tabs], [t[c] for t in tabs])
        losses[step] = float(out[0])
        if step % 20 == 0:
            print("step %d  %.0f s  total %.4f (reference %.4f)" % (step, time.time() - t0, losses[step], z["losses"][step, 0]), flush=True)
    P = {k: v.detach().numpy() for k, v in trn.P.items()}
    vec, _ = O.news_encoder_fwd(P, comb, nl, A)
    per = []
    for ln in z["test_lines"]:
        f = str(ln).split("\t")
        y = np.array([int(x.split("-")[1]) for x in f[4].split()])
        if y.mean() in (0, 1):
            continue
        h, m = DO.pad_to_fix_len(DO.trans_to_nindex(news_index, f[3].split()), U)
        cidx = DO.trans_to_nindex(news_index, [x.split("-")[0] for x in f[4].split()])
        u, _ = O.user_encoder_fwd(P, "student.user_encoder.", vec[np.array(h)][None], np.array(m, np.float32)[None], True)
        sc = vec[np.array(cidx)] @ u[0]
        per.append([PM.roc_auc_score(y, sc), PM.mrr_score(y, sc), PM.ndcg_score(y, sc, 5), PM.ndcg_score(y, sc, 10)])
    got = np.mean(per, 0)
    rec = {"what": "oracle/torch_port.py (fp32, torch CPU) trained %d steps on the quality golden's batches from the same init, evaluated by the "
                   "fp32 oracle; gap to the reference's own run" % steps,
           "metrics": dict(zip(("AUC", "MRR", "nDCG@5", "nDCG@10"), [round(float(x), 5) for x in got])),
           "reference": dict(zip(("AUC", "MRR", "nDCG@5", "nDCG@10"), [round(float(x), 5) for x in z["metrics_unquantised"]])),
           "gap_pt": [round(100 * float(a - b), 3) for a, b in zip(got, z["metrics_unquantised"])],
           "loss_err_first_50_max": float(np.abs(losses[:50] - z["losses"][:50, 0]).max()),
           "loss_err_median": float(np.median(np.abs(losses - z["losses"][:, 0]))),
           "loss_last20": [float(losses[-20:].mean()), float(z["losses"][-20:, 0].mean())]}
    print(json.dumps(rec))
    with open(os.path.join(ROOT, "profiles", "r06_quality_noise_floor.json"), "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
