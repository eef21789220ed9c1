"""Randomised differential test of one training step (HIP engine, fp16 build) against the numpy oracle over random
shapes, masks and options (development aid; run on the GPU box: python tools/fuzz_engine.py [n_cases] [seed])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import engine as E, hashinit
from schema import FULL, state_shapes
from oracle import newsrec_oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
DTYPE = os.environ.get("DTYPE", "fp16")
TOLS = dict(fp16=(2e-3, 3e-3, 2e-2), bf16=(1.6e-2, 2.4e-2, 8e-2))[DTYPE]
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    nl = int(rs.randint(1, 3))
    tr = tuple(sorted(rs.choice(nl, size=rs.randint(1, nl + 1), replace=False).tolist()))
    B, U, C, L = int(rs.randint(1, 7)), int(rs.randint(1, 51)), int(rs.randint(2, 7)), int(rs.randint(2, 33))
    if os.environ.get("LONG_TITLES"):          # stage-2 shapes on the long-sequence attention kernels
        L, U = int(rs.randint(33, 97)), int(rs.randint(1, 13))
    pooling = ["att", "att", "cls", "mean"][rs.randint(4)]
    nrms = bool(rs.rand() < 0.25)
    D = 256 if nrms else int(rs.choice([64, 128, 256]))
    Q = int(rs.choice([40, 100, 200]))
    T_ = int(rs.randint(1, 5))
    ulm = bool(rs.rand() < 0.5)
    tau, coef = float(rs.choice([1.0, 2.0])), float(rs.choice([0.2, 1.0]))
    dims = dict(FULL, Q=Q)
    P = hashinit.init_state_dict(1000 + case, state_shapes(dims, nl, D, T_, pooling, 16 if nrms else 0))
    cfg = dict(n_layers=nl, heads=12, trainable_layers=list(tr), user_log_mask=ulm, temperature=tau, coef=coef, pooling=pooling,
               nrms_heads=16 if nrms else 0)
    def toks(n):
        out = np.zeros((n, 2 * L), np.int64)
        for r in range(n):
            k = rs.randint(0, L + 1) if rs.rand() < 0.1 else rs.randint(1, L + 1)
            out[r, :k] = rs.randint(1, 30522, k); out[r, L:L + k] = 1
        return out
    hist, cand = toks(B * U).reshape(B, U, 2 * L), toks(B * C).reshape(B, C, 2 * L)
    mask = (rs.rand(B, U) > rs.rand()).astype(np.float32)
    label = rs.randint(0, C, B)
    th = [rs.randn(B, U, D).astype(np.float32) * 0.3 for _ in range(T_)]
    tc = [rs.randn(B, C, D).astype(np.float32) * 0.3 for _ in range(T_)]
    ec = E.EngineConfig(n_layers=nl, trainable_layers=tr, num_teachers=T_, user_log_length=U, npratio=C - 1, num_words=L, news_dim=D,
                        news_query=Q, user_query=Q, user_log_mask=ulm, temperature=tau, coef=coef, pooling=pooling,
                        nrms_heads=16 if nrms else 0)
    if os.environ.get("ONLY") and case != int(os.environ["ONLY"]):     # same random stream, one case (ONLY=35), e.g. with
        continue                                                         # USER_FUSED=0 / SG_KDIV=128 to bisect a failure
    eng = E.Engine(ec, "cuda:0", max_batch=B, dtype=DTYPE)
    if os.environ.get("USER_FUSED") == "0": eng.fused_user_fwd = False
    if os.environ.get("SG_KDIV"): eng.sg_kdiv = int(os.environ["SG_KDIV"])
    eng.load_state_dict(P)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    losses, score = eng.forward(t(hist), t(mask), t(cand), t(label), [t(x) for x in th], [t(x) for x in tc])
    eng.backward(); torch.cuda.synchronize()
    out = O.model_fwd(P, cfg, hist, mask, cand, label, th, tc); G = O.model_bwd(P, cfg, out)
    ref_l = np.array([out["distill_loss"], out["target_loss"], out["emb_loss"]])
    le = np.abs(losses[:3].cpu().numpy() - ref_l).max() / max(1.0, np.abs(ref_l).max())
    un = np.sqrt((out["user"] ** 2).sum(-1)).max()
    se = np.abs(score.cpu().numpy() - out["student_score"]).max() / max(1.0, np.abs(out["student_score"]).max(), un if nrms else 0)
    top = max(np.sqrt((g.astype(np.float64) ** 2).sum()) for g in G.values())
    worst, wk = 0.0, ""
    for k in eng.grads:
        if k.endswith("self.key.bias") or k.endswith("att_fc2.bias") or k.endswith("W_K.bias"):
            continue
        ref = G[k]; rn = np.sqrt((ref.astype(np.float64) ** 2).sum())
        if rn < 1e-4 * top: continue
        err = np.sqrt(((eng.grad(k).cpu().numpy() - ref).astype(np.float64) ** 2).sum()) / rn
        if err > worst: worst, wk = err, k
        if os.environ.get("ONLY") and err > 0.5 * TOLS[2]: print("   %-60s |ref| %.3e (top %.3e) err %.2e" % (k, rn, top, err))
    ok = le < TOLS[0] and se < TOLS[1] and worst < TOLS[2] and np.isfinite(le + se + worst)
    bad += not ok
    print("%s case %2d nl=%d tr=%s B=%d U=%d C=%d L=%d D=%d Q=%d T=%d ulm=%d pool=%s nrms=%d : loss %.1e score %.1e grad %.1e %s" % (
        "ok " if ok else "BAD", case, nl, tr, B, U, C, L, D, Q, T_, ulm, pooling, nrms, le, se, worst, "" if ok else wk), flush=True)
    del eng
print("failures:", bad)
sys.exit(1 if bad else 0)
