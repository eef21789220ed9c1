"""GPU parity of the stage-1 KD step (Post-train_KD.ipynb DistillModel) through the C ABI:
bodies of 128 tokens on the long-sequence attention kernels, titles on the L<=32 kernels, one shared parameter set.
Checked against the golden vectors captured from the notebook's own modules and against the numpy oracle.
Tolerances as in test_engine_gpu.py (fp16: the north-star 1e-3 * max(1,|ref|); bf16: x16)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import engine as E                           # noqa: E402
from helpers import load_stage1_case        # noqa: E402
from oracle import newsrec_oracle as O      # noqa: E402
from stage1 import Stage1Engine             # noqa: E402

DEV = "cuda:0"
TOL = {"bf16": 1.6e-2, "fp16": 1e-3}
GTOL = {"bf16": 6e-2, "fp16": 1.5e-2}


def _make(z, cfg, dtype):
    seed, B, T_, C, Lt, Lb, D, A, nl = [int(x) for x in z["meta"]]
    eng = Stage1Engine(n_layers=nl, trainable_layers=cfg["trainable_layers"], num_teachers=T_, npratio=C - 1, title_len=Lt,
                       body_len=Lb, device=DEV, batch=B, dtype=dtype, news_dim=D)
    return eng, B


def _dev(inp):
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    title, body, label, tt, tb = inp
    return t(title), t(body), t(label), [t(x) for x in tt], [t(x) for x in tb]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_stage1_step_matches_notebook_and_oracle(dtype):
    z, P, cfg, inp = load_stage1_case("stage1_full.npz")
    eng, B = _make(z, cfg, dtype)
    assert not any(k.startswith("teachers.") or "user_encoder" in k for k in eng.shapes)     # DistillModel's schema
    eng.load_state_dict(P)
    losses, score = eng.forward(*_dev(inp))
    torch.cuda.synchronize()
    l = losses.cpu().numpy()
    got = dict(distill=l[0], target=l[1], emb=l[2], total=float(eng.total_loss().item()))
    tol = TOL[dtype]
    sc = score.cpu().numpy()
    print("\n[stage1 %s] score max|err| %.3e (|ref| max %.2f)" % (dtype, np.abs(sc - z["score"]).max(), np.abs(z["score"]).max()))
    for k in got:
        print("   %s: got %.6f ref %.6f err %.2e" % (k, got[k], float(z[k]), abs(got[k] - float(z[k]))))
        assert abs(got[k] - float(z[k])) <= tol * max(1.0, abs(float(z[k]))), k
    assert np.abs(sc - z["score"]).max() <= tol * max(1.0, np.abs(z["score"]).max())

    out = O.distill_fwd(P, cfg, *inp)
    N = B * eng.cfg_t.C
    S = eng.title.S.cpu().numpy()
    np.testing.assert_allclose(S[:N].reshape(out["title_vec"].shape), out["title_vec"], rtol=0,
                               atol=tol * max(1.0, np.abs(out["title_vec"]).max()))
    np.testing.assert_allclose(S[N:N + B], out["body_vec"], rtol=0, atol=tol * max(1.0, np.abs(out["body_vec"]).max()))

    eng.backward()
    torch.cuda.synchronize()
    G = O.distill_bwd(P, cfg, out)
    worst = 0.0
    assert set(eng.title.grads) == set(G), set(eng.title.grads) ^ set(G)
    for k in eng.title.grads:
        ref = G[k]
        got_g = eng.grad(k).cpu().numpy()
        if k.endswith("self.key.bias") or k.endswith("att_fc2.bias"):
            assert np.abs(got_g).max() < 1e-3          # mathematical no-ops: rounding noise only
            continue
        rn = np.sqrt((ref.astype(np.float64) ** 2).sum())
        err = np.sqrt(((got_g - ref).astype(np.float64) ** 2).sum()) / (rn + 1e-12)
        worst = max(worst, err)
        assert err < GTOL[dtype], "%s: relative L2 error %.3e (norm %.3e)" % (k, err, rn)
        gn = float(z["gnorm." + k])                     # fp32 autograd of the notebook's DistillModel
        assert abs(np.sqrt((got_g.astype(np.float64) ** 2).sum()) - gn) <= GTOL[dtype] * gn + 1e-7, k
    print("   worst gradient relative L2 error %.3e" % worst)

    # the body pass accumulates into the title pass's gradients, the title pass overwrites: a repeated step
    # reproduces them bit for bit (fixed-order reductions, no atomics)
    g0 = eng.title.flat_g.clone()
    eng.forward(*_dev(inp))
    eng.backward()
    torch.cuda.synchronize()
    assert torch.equal(g0, eng.title.flat_g)


def test_stage1_loss_decreases():
    z, P, cfg, inp = load_stage1_case("stage1_full.npz")
    eng, B = _make(z, cfg, "bf16")
    eng.load_state_dict(P)
    d = _dev(inp)
    hist = []
    for _ in range(5):
        eng.forward(*d)
        hist.append(float(eng.total_loss().item()))
        eng.backward()
        eng.step(lr=1e-4)
    print("\nstage-1 loss trajectory:", ["%.4f" % x for x in hist])
    assert hist[-1] < hist[0]
