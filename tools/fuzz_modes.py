"""Randomised consistency test of the identical-result modes (development aid; GPU box):
  (a) indexed feed with in-batch de-duplication and / or the frozen-layer cache vs the plain indexed feed, over a sequence of
      steps with changing batch composition (different encoded-sequence counts step to step);
  (b) stage-0 / stage-1 steps at random title / body lengths vs the numpy oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import engine as E, hashinit, synth
from dedup import build_plan
from schema import FULL, state_shapes
from oracle import newsrec_oracle as O
from stage1 import Stage1Engine

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
# ---------------------------------------------------------------- (a)
for case in range(8):
    nl = int(rs.randint(2, 4)); lo = int(rs.randint(0, nl)); tr = tuple(range(lo, nl))
    B, U, C, L, T_ = int(rs.randint(1, 9)), int(rs.randint(2, 51)), int(rs.randint(2, 6)), int(rs.randint(4, 31)), int(rs.randint(1, 4))
    n = int(rs.randint(5, 120))
    D = 256
    P = hashinit.init_state_dict(2000 + case, state_shapes(FULL, nl, D, T_))
    ec = E.EngineConfig(n_layers=nl, trainable_layers=tr, num_teachers=T_, user_log_length=U, npratio=C - 1, num_words=L, news_dim=D,
                        user_log_mask=bool(rs.rand() < 0.5))
    comb = torch.from_numpy(synth.news_table(case, n, L)).cuda()
    tables = torch.from_numpy(synth.teacher_tables(case, T_, n, D)).cuda()
    res = {}
    for mode in ("plain", "dedup", "cache", "dedup+cache"):
        eng = E.Engine(ec, "cuda:0", max_batch=B, dtype="bf16"); eng.load_state_dict(P)
        if "cache" in mode: eng.build_frozen_cache(comb)
        r2 = np.random.RandomState(case)
        outs = []
        for step in range(4):
            h = r2.randint(0, n + 1, (B, U)).astype(np.int32); h[:, :r2.randint(0, U)] = 0
            m = (h > 0).astype(np.float32); c = r2.randint(1, n + 1, (B, C)).astype(np.int32); y = r2.randint(0, C, B)
            plan = build_plan(h, c) if "dedup" in mode else None
            t = lambda x: torch.from_numpy(x).cuda()
            l, s = eng.forward_indexed(comb, t(h), t(m), t(c), t(y), tables, plan.to("cuda") if plan is not None else None)
            eng.backward(); eng.step(1e-4)
            outs.append((l.clone(), s.clone()))
        torch.cuda.synchronize()
        res[mode] = (outs, eng.flat[True].clone())
    ref = res["plain"]
    for mode in ("dedup", "cache", "dedup+cache"):
        d_l = max(float((a[0] - b[0]).abs().max()) for a, b in zip(res[mode][0], ref[0]))
        d_p = float((res[mode][1] - ref[1]).abs().max())
        first_equal = torch.equal(res[mode][0][0][0], ref[0][0][0]) and torch.equal(res[mode][0][0][1], ref[0][0][1])
        exact = mode == "cache"
        # de-duplication changes gradient rounding; AMSGrad's normalised update turns a flipped tiny gradient into up to 2 * lr
        # of parameter difference per step (lr 1e-4, 4 steps)
        # (the loss bound is empirical for bf16 runs: 6.3e-3 seen at seed 5 with the parameter drift inside its bound)
        ok = first_equal and (d_l == 0.0 and d_p == 0.0 if exact else d_l < 1e-2 and d_p <= 8.5e-4)
        bad += not ok
        print("%s (a) case %d nl=%d tr=%s B=%d U=%d C=%d L=%d n=%d %-11s: first step identical %s, max loss drift over 4 steps %.1e, param drift %.1e" % (
            "ok " if ok else "BAD", case, nl, tr, B, U, C, L, n, mode, first_equal, d_l, d_p), flush=True)
# ---------------------------------------------------------------- (b)
for case in range(10):
    nl = int(rs.randint(1, 3)); tr = tuple(sorted(rs.choice(nl, size=rs.randint(1, nl + 1), replace=False).tolist()))
    B, C, Lt, Lb, T_ = int(rs.randint(1, 5)), int(rs.randint(2, 7)), int(rs.randint(3, 33)), int(rs.randint(33, 200)), int(rs.randint(0, 4))
    if case % 5 == 3:
        Lt = int(rs.randint(33, 90))          # round 6 (joint passes): both passes on the long-sequence kernels ...
    if case % 5 == 4:
        Lb = int(rs.randint(8, 33))           # ... or both on the L <= 32 kernels
    D = int(rs.choice([64, 256]))
    shapes = {k: v for k, v in state_shapes(FULL, nl, D, T_).items() if k.startswith("student.news_encoder.") or k.startswith("transform_matrix.")}
    P = hashinit.init_state_dict(3000 + case, shapes)
    def toks(nn, L):
        out = np.zeros((nn, 2 * L), np.int64)
        for r in range(nn):
            k = rs.randint(1, L + 1); out[r, :k] = rs.randint(1, 30522, k); out[r, L:L + k] = 1
        return out
    title, body = toks(B * C, Lt).reshape(B, C, 2 * Lt), toks(B, Lb)
    label = rs.randint(0, C, B)
    tt = [rs.randn(B, C, D).astype(np.float32) * 0.3 for _ in range(T_)]; tb = [rs.randn(B, D).astype(np.float32) * 0.3 for _ in range(T_)]
    eng = Stage1Engine(n_layers=nl, trainable_layers=tr, num_teachers=T_, npratio=C - 1, title_len=Lt, body_len=Lb, device="cuda:0",
                       batch=B, dtype="fp16", news_dim=D)
    eng.load_state_dict(P)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    losses, score = eng.forward(t(title), t(body), t(label), [t(x) for x in tt], [t(x) for x in tb])
    eng.backward(); torch.cuda.synchronize()
    cfg = dict(n_layers=nl, heads=12, trainable_layers=list(tr))
    out = O.distill_fwd(P, cfg, title, body, label, tt, tb); G = O.distill_bwd(P, cfg, out)
    le = abs(float(eng.total_loss().item()) - float(out["total_loss"])) / max(1.0, float(out["total_loss"]))
    se = np.abs(score.cpu().numpy() - out["student_score"]).max() / max(1.0, np.abs(out["student_score"]).max())
    top = max(np.sqrt((g.astype(np.float64) ** 2).sum()) for g in G.values())
    worst = 0.0
    for k in eng.title.grads:
        if k.endswith("self.key.bias") or k.endswith("att_fc2.bias"): continue
        rn = np.sqrt((G[k].astype(np.float64) ** 2).sum())
        if rn < 1e-4 * top: continue
        worst = max(worst, np.sqrt(((eng.grad(k).cpu().numpy() - G[k]).astype(np.float64) ** 2).sum()) / rn)
    ok = le < 2e-3 and se < 3e-3 and worst < 2e-2
    bad += not ok
    print("%s (b) case %d nl=%d tr=%s B=%d C=%d Lt=%d Lb=%d D=%d T=%d : loss %.1e score %.1e grad %.1e" % (
        "ok " if ok else "BAD", case, nl, tr, B, C, Lt, Lb, D, T_, le, se, worst), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
