"""GPU parity of the train-mode dropout path (the stage-0 / stage-1 notebooks call .train(): Post-train_KD.ipynb cell 19:6,
Domian-specific_Post-train.ipynb cell 16:6; sites tnlrv3/modeling.py:177, 224 and BertSelfOutput / BertOutput at :287, :306).
Masks are counter-based (csrc/dropout.h), so parity is "same bits -> same numbers":
  * the device generator against oracle/dropout_oracle.py, bit for bit, through both attention-probability accessors;
  * a stage-1 step under dropout against the golden captured from the notebook's own modules (their nn.Dropout forwards
    replaced by the same masks) and against the oracle's gradients;
  * p = 0 is bit-identical to the entry points without a dropout site; masks are regenerated identically in the backward pass
    (a repeated step reproduces the gradients bit for bit) and differ from call to call."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tnr_hip as T                            # noqa: E402
from helpers import load_stage1_case          # noqa: E402
from oracle import dropout_oracle as DO       # noqa: E402
from oracle import newsrec_oracle as O        # noqa: E402
from stage1 import Stage1Engine               # noqa: E402

DEV = "cuda:0"
TOL = {"bf16": 1.6e-2, "fp16": 1e-3}
GTOL = {"bf16": 6e-2, "fp16": 1.5e-2}


def test_device_masks_equal_the_oracle_bit_for_bit():
    for (p, seed, kind, layer, call, rows, cols) in ((0.1, 777, DO.KIND_FFN_OUT, 1, 0, 1000, 768), (0.25, 2 ** 40 + 5, DO.KIND_EMB, 0, 9, 333, 256),
                                                     (0.1, 1, DO.KIND_ATTN_OUT, 11, 2 ** 31, 64, 3072)):
        d = T.Dropout.site_of(p, seed, kind, layer, call)
        out = torch.empty((rows, cols), device=DEV)
        T.call("tnr_dropout_mask", d, rows, cols, out)
        want = DO.rows_mask(p, seed, DO.site_id(kind, layer), call, rows, cols)
        assert np.array_equal(out.cpu().numpy(), want)
        assert abs(float((out == 0).float().mean()) - DO.threshold(p) / 65536.0) < 4e-3
    for (p, seed, layer, call, n_seq, A, L) in ((0.1, 777, 0, 1, 5, 12, 30), (0.1, 3, 2, 4, 2, 12, 128), (0.3, 9, 1, 0, 3, 4, 37)):
        d = T.Dropout.site_of(p, seed, DO.KIND_PROB, layer, call)
        want = DO.probs_mask(p, seed, DO.site_id(DO.KIND_PROB, layer), call, n_seq, A, L).reshape(n_seq * A, L, L)
        for by_cols in (0, 1):                 # per-query accessor (forward, dQ) and per-key accessor (dK / dV)
            out = torch.full((n_seq * A, L, L), -1.0, device=DEV)
            T.call("tnr_dropout_mask_probs", d, n_seq * A, L, by_cols, out)
            assert np.array_equal(out.cpu().numpy(), want), (L, by_cols)


def _make(z, cfg, dtype):
    seed, B, T_, C, Lt, Lb, D, A, nl = [int(x) for x in z["meta"]]
    eng = Stage1Engine(n_layers=nl, trainable_layers=cfg["trainable_layers"], num_teachers=T_, npratio=C - 1, title_len=Lt,
                       body_len=Lb, device=DEV, batch=B, dtype=dtype, news_dim=D)
    return eng, B


def _dev(inp):
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    title, body, label, tt, tb = inp
    return t(title), t(body), t(label), [t(x) for x in tt], [t(x) for x in tb]


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_stage1_train_mode_step_matches_notebook_and_oracle(dtype):
    """BASELINE configs[4] shapes (2 layers, 4 teachers, 1+4 titles of 30 tokens on the L <= 32 kernels, bodies of 128 on the long
    ones) under hidden / attention-probability dropout 0.1 / 0.1."""
    z, P, cfg, inp = load_stage1_case("stage1_cfg4_drop.npz")
    p_h, p_a, seed = float(z["dropout"][0]), float(z["dropout"][1]), int(z["dropout"][2])
    eng, B = _make(z, cfg, dtype)
    eng.load_state_dict(P)
    eng.set_dropout(p_h, p_a, seed)
    losses, score = eng.forward(*_dev(inp))
    torch.cuda.synchronize()
    l = losses.cpu().numpy()
    got = dict(distill=l[0], target=l[1], emb=l[2], total=float(eng.total_loss().item()))
    tol = TOL[dtype]
    sc = score.cpu().numpy()
    print("\n[stage1 dropout %s] score max|err| %.3e (|ref| max %.2f)" % (dtype, np.abs(sc - z["score"]).max(), np.abs(z["score"]).max()))
    for k in got:
        print("   %s: got %.6f ref %.6f err %.2e" % (k, got[k], float(z[k]), abs(got[k] - float(z[k]))))
        assert abs(got[k] - float(z[k])) <= tol * max(1.0, abs(float(z[k]))), k
    assert np.abs(sc - z["score"]).max() <= tol * max(1.0, np.abs(z["score"]).max())

    out = O.distill_fwd(P, cfg, *inp, drop_title=DO.Dropout(p_h, p_a, seed, 0), drop_body=DO.Dropout(p_h, p_a, seed, 1))
    N = B * eng.cfg_t.C
    S = eng.title.S.cpu().numpy()
    np.testing.assert_allclose(S[:N].reshape(out["title_vec"].shape), out["title_vec"], rtol=0, atol=tol * max(1.0, np.abs(out["title_vec"]).max()))
    np.testing.assert_allclose(S[N:N + B], out["body_vec"], rtol=0, atol=tol * max(1.0, np.abs(out["body_vec"]).max()))

    eng.backward()
    torch.cuda.synchronize()
    G = O.distill_bwd(P, cfg, out)
    worst = 0.0
    for k in eng.title.grads:
        ref, got_g = G[k], eng.grad(k).cpu().numpy()
        if k.endswith("self.key.bias") or k.endswith("att_fc2.bias"):
            assert np.abs(got_g).max() < 1e-3
            continue
        rn = np.sqrt((ref.astype(np.float64) ** 2).sum())
        err = np.sqrt(((got_g - ref).astype(np.float64) ** 2).sum()) / (rn + 1e-12)
        worst = max(worst, err)
        assert err < GTOL[dtype], "%s: relative L2 error %.3e (norm %.3e)" % (k, err, rn)
        gn = float(z["gnorm." + k])                     # fp32 autograd of the notebook's DistillModel under the same masks
        assert abs(np.sqrt((got_g.astype(np.float64) ** 2).sum()) - gn) <= GTOL[dtype] * gn + 1e-7, k
    print("   worst gradient relative L2 error %.3e" % worst)

    # the next forward call draws new masks ...
    g0, s0 = eng.title.flat_g.clone(), score.clone()
    _, s1 = eng.forward(*_dev(inp))
    assert float((s1 - s0).abs().max()) > 1e-4
    # ... and rewinding the call counter reproduces the first step bit for bit (masks regenerated, not stored)
    eng.title.drop_calls = eng.body.drop_calls = 0
    _, s2 = eng.forward(*_dev(inp))
    eng.backward()
    torch.cuda.synchronize()
    assert torch.equal(s2, s0) and torch.equal(eng.title.flat_g, g0)


def test_dropout_p_zero_is_the_eval_path_bit_for_bit():
    z, P, cfg, inp = load_stage1_case("stage1_cfg4.npz")
    eng, B = _make(z, cfg, "fp16")
    eng.joint = False            # dropout runs the per-pass form (its masks are numbered per pass): compare like with like
    eng.load_state_dict(P)
    l0, s0 = eng.forward(*_dev(inp))
    l0, s0 = l0.clone(), s0.clone()
    eng.backward()
    g0 = eng.title.flat_g.clone()
    eng.set_dropout(0.0, 0.0, 5)
    assert eng.title.drop is None and eng.body.drop is None
    # and through the *_do entry points themselves: a site with p = 0 takes the same kernels with the mask switched off
    eng.set_dropout(0.1, 0.1, 5)
    for e in (eng.title, eng.body):
        e.drop["p_hidden"], e.drop["p_attn"] = 0.0, 0.0
    import tnr_hip
    orig = tnr_hip.Dropout.site_of
    try:
        tnr_hip.Dropout.site_of = classmethod(lambda cls, p, seed, kind, layer, call: cls(int(seed), int(kind) | (int(layer) << 8), int(call), 0.0))
        l1, s1 = eng.forward(*_dev(inp))
        eng.backward()
        torch.cuda.synchronize()
    finally:
        tnr_hip.Dropout.site_of = orig
    assert torch.equal(l0, l1) and torch.equal(s0, s1) and torch.equal(g0, eng.title.flat_g)
