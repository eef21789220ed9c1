"""GPU box: the ENGINE trained on the learnable corpus of tests/golden/quality_corpus.py and evaluated through the product's own
eval path (Engine.encode_news / user_vectors, metrics.py) - used to choose the corpus / learning rate / step count of the quality
golden BEFORE spending half an hour of CPU on the reference's run (make_golden.golden_quality), and to print the learning curve.
    python tools/quality_proto.py [lr=1e-4] [steps=200] [B=8] [every=50] [dtype=fp16]"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tiny-newsrec_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import engine as E                           # noqa: E402
import hashinit                              # noqa: E402
import metrics                               # noqa: E402
from decode_worker import decode_lines       # noqa: E402
from quality_corpus import quality_corpus    # noqa: E402
from schema import FULL, state_shapes        # noqa: E402


def evaluate(eng, dev_news, news_index, test_lines, U, B):
    ns = eng.encode_news(dev_news)
    host = ns.cpu().numpy()
    out = []
    for i in range(0, len(test_lines), B):
        rows = [l.split("\t") for l in test_lines[i:i + B]]
        h = np.zeros((len(rows), U), np.int32)
        m = np.zeros((len(rows), U), np.float32)
        for r, f in enumerate(rows):
            click = [news_index.get(x, 0) for x in f[3].split()][-U:]
            if click:
                h[r, U - len(click):] = click
                m[r, U - len(click):] = 1
        uv = eng.user_vectors(ns, torch.from_numpy(h).cuda(), torch.from_numpy(m).cuda()).cpu().numpy()
        for r, f in enumerate(rows):
            c = [news_index.get(x.split("-")[0], 0) for x in f[4].split()]
            y = np.array([int(x.split("-")[1]) for x in f[4].split()])
            if y.mean() in (0, 1):
                continue
            sc = host[c] @ uv[r]
            out.append([metrics.roc_auc_score(y, sc), metrics.mrr_score(y, sc), metrics.ndcg_score(y, sc, 5), metrics.ndcg_score(y, sc, 10)])
    return np.mean(out, 0)


def main():
    kw = dict(lr=1e-4, steps=200, B=8, every=50, dtype="fp16", seed=71, T=2)
    for a in sys.argv[1:]:
        k, v = a.split("=")
        kw[k] = type(kw[k])(v)
    c = quality_corpus(kw["seed"], T=kw["T"])
    nl, T_, B, U, C = 2, kw["T"], kw["B"], 50, 5
    cfg = E.EngineConfig(n_layers=nl, trainable_layers=(0, 1), num_teachers=T_, user_log_mask=True, temperature=1.0, coef=0.2)
    eng = E.Engine(cfg, "cuda:0", max_batch=B, dtype=kw["dtype"])
    eng.load_state_dict(hashinit.init_state_dict(kw["seed"], state_shapes(FULL, nl, 256, T_)))
    dev_news = torch.from_numpy(c["news_combined"].astype(np.int32)).cuda()
    tabs = torch.from_numpy(np.stack(c["tables"], 0)).cuda()
    random.seed(kw["seed"])
    lines = [l.encode() for l in c["train_lines"]]
    print("step  total distill emb target | AUC MRR nDCG5 nDCG10", flush=True)
    hist = []
    for step in range(kw["steps"] + 1):
        if step % kw["every"] == 0:
            ev = evaluate(eng, dev_news, c["news_index"], c["test_lines"], U, B)
            print("%4d  %s | %s" % (step, np.round(np.mean(hist[-20:], 0), 4) if hist else "-", np.round(ev, 4)), flush=True)
        if step == kw["steps"]:
            break
        s0 = (step * B) % (len(lines) - B + 1)
        h, m, cc, y = decode_lines(lines[s0:s0 + B], c["news_index"], U, C - 1)
        t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x.astype(dt))).cuda()
        losses, score = eng.forward_indexed(dev_news, t(h, np.int32), t(m, np.float32), t(cc, np.int32), t(y, np.int64), tabs)
        eng.backward()
        eng.step(kw["lr"])
        l = losses.cpu().numpy()
        hist.append([l[0] + 0.2 * l[1] + l[2], l[0], l[2], l[1]])
    if eng.scaler.enabled:
        eng.scaler.drain(eng)
        print("fp16 steps skipped:", eng.scaler.skipped)


if __name__ == "__main__":
    main()
