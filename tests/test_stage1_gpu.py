"""GPU parity of the stage-1 KD step (Post-train_KD.ipynb DistillModel) through the C ABI:
bodies of 128 tokens on the long-sequence attention kernels, titles on the L<=32 kernels, one shared parameter set.
Checked against the golden vectors captured from the notebook's own modules and against the numpy oracle.
Tolerances as in test_engine_gpu.py (fp16: the north-star 1e-3 * max(1,|ref|); bf16: x16)."""
import os

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
@pytest.mark.parametrize("case", ["stage1_full.npz",        # notebook shapes scaled down: 2 teachers, 1+3 titles of 24, bodies of 128
                                  "stage1_cfg4.npz"])       # BASELINE configs[4] exactly: 4 teachers, 1+4 titles of 30, bodies of 128
def test_stage1_step_matches_notebook_and_oracle(dtype, case):
    z, P, cfg, inp = load_stage1_case(case)
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


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_stage1_chained_weight_gradients_equal_the_two_pass_form(dtype):
    """Stage1Engine.backward runs the title and the body backward in step and launches every shared weight's gradient as one
    chained problem (tnr_gemm_tn_wgrad_group, accumulate = 2).  Against chain_wgrad = False (title writes, body adds): everything
    that is not a weight gradient bit for bit; weight gradients the same products summed in another association (fp32 rounding).
    Under a bucket hook (data parallelism) the hooks fire in the same order, each after both passes' contributions are out."""
    z, P, cfg, inp = load_stage1_case("stage1_cfg4.npz")
    eng, B = _make(z, cfg, dtype)
    eng.load_state_dict(P)
    d = _dev(inp)
    res = {}
    eng.joint = False                                    # the per-pass form (round 5); the joint passes have their own test below
    for mode in ("two", "chain", "chain_hooked", "two_hooked", "chain_streams"):
        eng.chain_wgrad = mode.startswith("chain")
        eng.two_streams = mode.endswith("streams")       # the body pass on a second stream: the same kernels on the same operands
        fired = []
        eng.title.flat_g.fill_(float("nan"))
        eng.forward(*d)
        eng.backward(after_bucket=(lambda i: fired.append(i)) if mode.endswith("hooked") else None)
        torch.cuda.synchronize()
        res[mode] = (eng.title.flat_g.clone(), fired, eng.title.losses.clone(), eng.title.S.clone())
    assert res["chain_hooked"][1] == res["two_hooked"][1] and len(res["two_hooked"][1]) == 1 + 2 * len(cfg["trainable_layers"])
    t = eng.title
    for k in (0, 2, 3):
        assert torch.equal(torch.nan_to_num(res["chain"][k]), torch.nan_to_num(res["chain_streams"][k])), k     # (flat_g's padding stays NaN)
    for a, b_ in (("chain", "two"), ("chain_hooked", "two_hooked"), ("chain", "chain_hooked")):
        ga, gb = res[a][0], res[b_][0]
        for k, gk in t.grads.items():
            o = gk.storage_offset() - t.flat_g.storage_offset()
            va, vb = ga[o:o + gk.numel()], gb[o:o + gk.numel()]
            assert bool(torch.isfinite(va).all()), (a, k)          # every gradient was written (flat_g was all NaN)
            if torch.equal(va, vb):
                continue
            is_wgrad = k.endswith(".weight") and gk.dim() == 2 and ("encoder.layer" in k or k.endswith("attn.att_fc1.weight"))
            assert is_wgrad and (a, b_) != ("chain", "chain_hooked"), (a, b_, k)
            assert float((va - vb).abs().max()) <= 4e-6 * float(vb.abs().max()) + 1e-12, (a, b_, k)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("case", ["stage1_cfg4.npz", "stage1_full.npz"])
def test_stage1_joint_passes_equal_the_per_pass_form(dtype, case):
    """Round 6: the body pass's token rows directly behind the title pass's, every Linear / LayerNorm / weight gradient ONE launch
    over both (Stage1Engine.joint).  Against the per-pass form (joint = False: a launch per pass, chained weight gradients): news
    vectors, scores and all three losses BIT FOR BIT (a row's K order does not depend on the launch it is in); parameter gradients
    the same sums in another order (fp32 rounding); under a bucket hook the hooks fire in the same order; one stream == two
    streams bit for bit."""
    z, P, cfg, inp = load_stage1_case(case)
    eng, B = _make(z, cfg, dtype)
    eng.load_state_dict(P)
    d = _dev(inp)
    res = {}
    for mode in ("per_pass", "joint", "joint_hooked", "per_pass_hooked", "joint_one_stream"):
        eng.joint = mode.startswith("joint")
        eng.joint_streams = mode != "joint_one_stream"
        fired = []
        eng.title.flat_g.fill_(float("nan"))
        eng.title.S.fill_(float("nan"))
        eng.forward(*d)
        assert eng.ran_joint == eng.joint
        eng.backward(after_bucket=(lambda i: fired.append(i)) if mode.endswith("hooked") else None)
        torch.cuda.synchronize()
        Rt = eng.cur[2]
        res[mode] = (eng.title.flat_g.clone(), fired, eng.title.losses.clone(), eng.title.S[:Rt].clone(), eng.title.score[:B].clone())
    assert res["joint_hooked"][1] == res["per_pass_hooked"][1] and len(res["joint_hooked"][1]) == 1 + 2 * len(cfg["trainable_layers"])
    for k in (2, 3, 4):
        assert bool(torch.isfinite(res["joint"][k]).all())
        assert torch.equal(res["joint"][k], res["per_pass"][k]), k
    for k in (0, 2, 3, 4):
        assert torch.equal(torch.nan_to_num(res["joint"][k]), torch.nan_to_num(res["joint_one_stream"][k])), k
    for k in (2, 3, 4):
        assert torch.equal(res["joint"][k], res["joint_hooked"][k]), k
    t = eng.title
    worst = 0.0
    # (under a bucket hook a layer's weight gradients leave in two grouped launches of two instead of one of four: other splits)
    for other in ("per_pass", "joint_hooked"):
        for k, gk in t.grads.items():
            o = gk.storage_offset() - t.flat_g.storage_offset()
            va, vb = res["joint"][0][o:o + gk.numel()], res[other][0][o:o + gk.numel()]
            assert bool(torch.isfinite(va).all()), k                   # every gradient was written (flat_g was all NaN)
            e = float((va - vb).norm()) / max(float(vb.norm()), 1e-30)
            worst = max(worst, e)
            assert e <= 2e-5, (other, k, e)
    print("\n[stage1 joint %s %s] worst relative L2 gap of a parameter gradient to the per-pass form: %.2e" % (case, dtype, worst))


def test_stage1_per_pass_step_after_joint_steps_sees_a_clean_workspace():
    """The joint passes put body rows where the title engine's own kernels expect zero rows (the weight gradient reads up to the
    next multiple of 64 behind the N Lt title rows): a per-pass step that follows joint ones (dropout switched on mid-run, a tools/
    A/B) must not see them.  B = 3 of the golden batch: 450 title rows, not a multiple of 64 - every gradient bit for bit what a
    fresh engine's per-pass step gives."""
    z, P, cfg, inp = load_stage1_case("stage1_cfg4.npz")
    d = _dev(inp)
    sub = (d[0][:3], d[1][:3], d[2][:3], [x[:3] for x in d[3]], [x[:3] for x in d[4]])
    eng, B = _make(z, cfg, "fp16")
    eng.load_state_dict(P)
    for _ in range(2):
        eng.forward(*sub)
        assert eng.ran_joint
        eng.backward()
    eng.joint = False
    eng.forward(*sub)
    assert not eng.ran_joint
    eng.backward()
    torch.cuda.synchronize()
    g1, l1 = eng.title.flat_g.clone(), eng.title.losses.clone()
    fresh, _ = _make(z, cfg, "fp16")
    fresh.joint = False
    fresh.load_state_dict(P)
    fresh.forward(*sub)
    fresh.backward()
    torch.cuda.synchronize()
    assert torch.equal(l1, fresh.title.losses) and torch.equal(g1, fresh.title.flat_g)


def test_stage1_joint_training_follows_the_per_pass_form():
    """Eight optimiser steps of the joint passes against the per-pass form from the same start (no dropout): the first step's
    losses bit for bit, every later loss within 1e-3 (the gradients differ by fp32 summation order only - 2e-5 relative, the test
    above - and Adam turns a gradient at the rounding floor into a +-lr step: measured 1.8e-4 after eight steps), and a short last
    batch (the body rows then sit behind FEWER title rows: the workspace is laid out again) runs and matches too."""
    z, P, cfg, inp = load_stage1_case("stage1_cfg4.npz")
    d = _dev(inp)
    res = []
    for joint in (False, True):
        eng, B = _make(z, cfg, "fp16")
        eng.joint = joint
        eng.load_state_dict(P)
        ls = []
        for i in range(8):
            eng.forward(*d)
            ls.append(eng.title.losses.clone())
            eng.backward()
            eng.step(1e-4, lr_bert=1e-5)
        half = (d[0][:B // 2], d[1][:B // 2], d[2][:B // 2], [x[:B // 2] for x in d[3]], [x[:B // 2] for x in d[4]])
        eng.forward(*half)
        assert eng.ran_joint == joint
        ls.append(eng.title.losses.clone())
        eng.backward()
        eng.step(1e-4, lr_bert=1e-5)
        eng.forward(*d)
        ls.append(eng.title.losses.clone())
        torch.cuda.synchronize()
        res.append(torch.stack(ls).cpu().numpy())
    assert np.array_equal(res[0][0], res[1][0])
    assert np.isfinite(res[1]).all()
    np.testing.assert_allclose(res[1], res[0], rtol=0, atol=1e-3)


def test_stage1_training_on_two_streams_equals_one_stream_bit_for_bit():
    """Eight optimiser steps (dropout live, the fp16 loss scaler live) with the body pass on its own stream against the same steps
    on one stream: parameters, optimiser state and every loss equal bit for bit - the cross-stream joins order every shared
    buffer (gradients, the chained weight-gradient slabs, the student rows) or the two runs would drift apart."""
    z, P, cfg, inp = load_stage1_case("stage1_cfg4.npz")
    d = _dev(inp)
    res = []
    for streams in (False, True, True):
        eng, B = _make(z, cfg, "fp16")
        eng.two_streams = streams
        eng.load_state_dict(P)
        eng.set_dropout(0.1, 0.1, seed=3)
        losses = []
        for _ in range(8):
            l, _s = eng.forward(*d)
            losses.append(l.clone())
            eng.backward()
            eng.step(1e-4, lr_bert=1e-5)
        torch.cuda.synchronize()
        t = eng.title
        res.append((torch.stack(losses), t.flat[True].clone(), t.adam_m.clone(), t.adam_v.clone()))
        del eng
    for other in res[1:]:
        for a, b_ in zip(res[0], other):
            assert torch.equal(a, b_)
    assert float(res[0][0][-1, 1]) < float(res[0][0][0, 1])         # and it trains


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


def test_stage1_indexed_feed_and_plain_adam():
    """forward_indexed (resident token / teacher tables + B x (1+K) document indices, the notebook's DistillDataset at
    index level) equals the materialised feed bit for bit; step(amsgrad=False) is torch's plain Adam with two rates."""
    z, P, cfg, inp = load_stage1_case("stage1_full.npz")
    eng, B = _make(z, cfg, "fp16")
    eng.load_state_dict(P)
    title, body, label, tt, tb = _dev(inp)
    C, D = eng.cfg_t.C, eng.cfg_t.D
    l0, s0 = eng.forward(title, body, label, tt, tb)
    l0, s0 = l0.clone(), s0.clone()
    eng.backward()
    g0 = eng.title.flat_g.clone()
    # build tables whose rows are this batch's documents (+ distractor rows) and the index batch that selects them
    n = B * C + 3
    rs = np.random.RandomState(0)
    perm = rs.permutation(n)[:B * C].reshape(B, C)
    t_tab = torch.zeros((n, title.shape[2]), dtype=torch.int32, device=DEV)
    b_tab = torch.zeros((n, body.shape[1]), dtype=torch.int32, device=DEV)
    tt_tab = torch.zeros((len(tt), n, D), device=DEV)
    tb_tab = torch.zeros((len(tt), n, D), device=DEV)
    pidx = torch.from_numpy(perm).to(DEV)
    t_tab[pidx.reshape(-1)] = title.reshape(B * C, -1).to(torch.int32)
    b_tab[pidx[:, 0]] = body.to(torch.int32)
    for i in range(len(tt)):
        tt_tab[i, pidx.reshape(-1)] = tt[i].reshape(B * C, D)
        tb_tab[i, pidx[:, 0]] = tb[i]
    l1, s1 = eng.forward_indexed(t_tab, b_tab, pidx.to(torch.int32), label, tt_tab, tb_tab)
    eng.backward()
    torch.cuda.synchronize()
    assert torch.equal(l0, l1) and torch.equal(s0, s1) and torch.equal(g0, eng.title.flat_g)
    # two-rate plain Adam against torch.optim.Adam on a bert parameter and a head parameter
    kb = E.layer_param_order(1)[12]                       # layer-1 output.dense.weight (bert_model group)
    kh = "transform_matrix.0.weight"
    ref = {}
    for k, lr in ((kb, 1e-6), (kh, 1e-5)):
        p = torch.nn.Parameter(eng.title.params[k].detach().cpu().clone())
        ref[k] = (p, torch.optim.Adam([p], lr=lr), lr)
    for step in range(2):
        eng.forward_indexed(t_tab, b_tab, pidx.to(torch.int32), label, tt_tab, tb_tab)
        eng.backward()
        for k, (p, opt, lr) in ref.items():
            p.grad = eng.grad(k).detach().cpu().clone()
            opt.step()
        eng.step(1e-5, lr_bert=1e-6)
    torch.cuda.synchronize()
    for k, (p, opt, lr) in ref.items():
        np.testing.assert_allclose(eng.title.params[k].cpu().numpy(), p.detach().numpy(), rtol=0, atol=0.02 * lr, err_msg=k)
    assert float(eng.title.adam_vmax.abs().max()) == 0.0        # the AMSGrad state is untouched


def test_post_train_kd_script_runs(tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "tiny-newsrec_amd"))
    cmd = [sys.executable, "-u", os.path.join(root, "tiny-newsrec_amd", "post_train_kd.py"), "--synthetic", "True", "--enable_hvd",
           "False", "--max_steps", "4", "--log_steps", "2", "--num_hidden_layers", "2", "--bert_trainable_layer", "0", "1",
           "--num_teachers", "2", "--npratio", "3", "--batch_size", "4", "--max_body_len", "128", "--synthetic_docs", "300",
           "--save_dir", str(tmp_path), "--dtype", "fp16"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=os.path.join(root, "tiny-newsrec_amd"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "d_loss" in (r.stdout + r.stderr) and "nan" not in (r.stdout + r.stderr).lower()
    sd = torch.load(str(tmp_path / "first_stage_2_layer.pt"), map_location="cpu")["model_state_dict"]
    assert "student.news_encoder.dense.weight" in sd and "transform_matrix.1.bias" in sd
    assert not any(k.startswith("teachers.") or "user_encoder" in k for k in sd)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_stage0_contrastive_step_matches_notebook(dtype):
    """Domian-specific_Post-train.ipynb TitleBodySimModel (12 layers, train 9-11, CE only) = Stage1Engine(num_teachers=0):
    against the golden captured by executing the notebook's cells 10-11 and the oracle's gradients."""
    from helpers import load_stage0_case
    z, P, cfg, inp = load_stage0_case()
    seed, B, T_, C, Lt, Lb, D, A, nl = [int(x) for x in z["meta"]]
    eng = Stage1Engine(n_layers=nl, trainable_layers=cfg["trainable_layers"], num_teachers=0, npratio=C - 1, title_len=Lt,
                       body_len=Lb, device=DEV, batch=B, dtype=dtype, news_dim=D)
    assert not any(k.startswith("transform_matrix") for k in eng.shapes)
    eng.load_state_dict(P)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    losses, score = eng.forward(t(inp[0]), t(inp[1]), t(inp[2]), [], [])
    torch.cuda.synchronize()
    tol = TOL[dtype]
    total = float(eng.total_loss().item())
    print("\n[stage0 %s] loss %.6f ref %.6f ; score max|err| %.2e (|ref| max %.2f)" % (
        dtype, total, float(z["total"]), np.abs(score.cpu().numpy() - z["score"]).max(), np.abs(z["score"]).max()))
    assert float(losses[0]) == 0.0 and float(losses[2]) == 0.0
    assert abs(total - float(z["total"])) <= tol * max(1.0, float(z["total"]))
    assert np.abs(score.cpu().numpy() - z["score"]).max() <= tol * max(1.0, np.abs(z["score"]).max())
    eng.backward()
    torch.cuda.synchronize()
    out = O.distill_fwd(P, cfg, *inp)
    G = O.distill_bwd(P, cfg, out)
    top = max(np.sqrt((g.astype(np.float64) ** 2).sum()) for g in G.values())
    worst = 0.0
    for k in eng.title.grads:
        ref = G[k]
        rn = np.sqrt((ref.astype(np.float64) ** 2).sum())
        if rn < 1e-4 * top:          # q / k / pooling-head gradients: at the noise floor of this 12-layer hash model
            continue
        err = np.sqrt(((eng.grad(k).cpu().numpy() - ref).astype(np.float64) ** 2).sum()) / rn
        worst = max(worst, err)
        assert err < GTOL[dtype], "%s: %.3e" % (k, err)
    print("   worst gradient relative L2 error %.3e" % worst)


def test_stage0_script_trains_and_exports(tmp_path):
    import pickle
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "tiny-newsrec_amd"))
    base = [sys.executable, "-u", os.path.join(root, "tiny-newsrec_amd", "post_train_kd.py"), "--stage", "0", "--synthetic", "True",
            "--enable_hvd", "False", "--num_hidden_layers", "3", "--bert_trainable_layer", "1", "2", "--npratio", "3",
            "--batch_size", "4", "--max_body_len", "96", "--synthetic_docs", "200", "--save_dir", str(tmp_path), "--dtype", "fp16"]
    r = subprocess.run(base + ["--max_steps", "4", "--save_steps", "2", "--log_steps", "2"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=os.path.join(root, "tiny-newsrec_amd"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for name in ("DP_3_layer_2.pt", "DP_3_layer_4.pt", "DP_3_layer.pt"):
        sd = torch.load(str(tmp_path / name), map_location="cpu")["model_state_dict"]
        assert "news_encoder.dense.weight" in sd and not any(k.startswith("student.") or k.startswith("transform") for k in sd)
    r = subprocess.run(base + ["--mode", "export", "--ckpt_paths", str(tmp_path / "DP_3_layer.pt"), str(tmp_path / "DP_3_layer_2.pt")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.join(root, "tiny-newsrec_amd"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for i in range(2):
        for which in ("title", "body"):
            emb = pickle.load(open(str(tmp_path / ("teacher_%s_emb_%d.pkl" % (which, i))), "rb"))
            assert emb.shape == (200, 256) and emb.dtype == np.float32 and np.isfinite(emb).all()
    a = pickle.load(open(str(tmp_path / "teacher_title_emb_0.pkl"), "rb"))
    b = pickle.load(open(str(tmp_path / "teacher_title_emb_1.pkl"), "rb"))
    assert not np.array_equal(a, b)                       # two different checkpoints = two different teachers


@pytest.mark.parametrize("Lt,Lb,Kn", [(30, 128, 4), (24, 512, 9)])       # bench.py's two stage-1 legs (the second: the notebook's own shape)
def test_stage1_bench_shape_step_matches_oracle(Lt, Lb, Kn):
    """bench.py's "configs[4] stage 1" leg at its OWN size (B = 32, 2-layer student training both layers, 4 teachers, 1 + 4 titles of
    30 tokens, bodies of 128: the joint passes over M = 8 896 token rows - the persistent 224 / 192-row tile plans, the grouped weight
    gradients sharing one round, the long-sequence kernels on 32 bodies) against the numpy oracle on the same synthetic inputs: the
    three losses and the (32, 5) scores at 1e-3, EVERY parameter gradient at the suite's fp16 bound (1.5e-2 relative L2)."""
    import hashinit
    import synth
    nl, T_, B, nd = 2, 4, 32, 3000
    eng = Stage1Engine(n_layers=nl, trainable_layers=(0, 1), num_teachers=T_, npratio=Kn, title_len=Lt, body_len=Lb, device=DEV, batch=B,
                       dtype="fp16")
    P = {k: hashinit.init_tensor(1234, k, tuple(sh)) for k, sh in eng.shapes.items()}
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    d_title = synth.news_table(11, nd - 1, Lt)
    d_body = synth.news_table(12, nd - 1, Lb, mean_len=0.6 * Lb, std_len=0.25 * Lb)
    d_tt = np.ascontiguousarray(synth.teacher_tables(13, T_, nd - 1, eng.cfg_t.D))
    d_tb = np.ascontiguousarray(synth.teacher_tables(14, T_, nd - 1, eng.cfg_t.D))
    rs = np.random.RandomState(1234)
    pidx = rs.randint(1, nd, (B, 1 + Kn)).astype(np.int32)
    label = rs.randint(0, 1 + Kn, B).astype(np.int64)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    losses, score = eng.forward_indexed(t(d_title), t(d_body), t(pidx), t(label), t(d_tt), t(d_tb))
    assert eng.ran_joint
    eng.backward()
    torch.cuda.synchronize()
    title = d_title.astype(np.int64)[pidx]                                # (B, 1 + K, 2 Lt) [ids | mask]
    body = d_body.astype(np.int64)[pidx[:, 0]]
    tt = [d_tt[i][pidx] for i in range(T_)]
    tb = [d_tb[i][pidx[:, 0]] for i in range(T_)]
    cfg = dict(n_layers=nl, heads=12, trainable_layers=[0, 1])
    out = O.distill_fwd(P, cfg, title, body, label, tt, tb)
    G = O.distill_bwd(P, cfg, out)
    le = abs(float(eng.total_loss().item()) - float(out["total_loss"])) / max(1.0, float(out["total_loss"]))
    se = np.abs(score.cpu().numpy() - out["student_score"]).max() / max(1.0, np.abs(out["student_score"]).max())
    nrm = lambda a: float(np.sqrt((a.astype(np.float64) ** 2).sum()))
    worst, worst_k = 0.0, None
    for k in eng.title.grads:
        got, ref = eng.grad(k).cpu().numpy(), G[k]
        if k.endswith("self.key.bias") or k.endswith("att_fc2.bias"):
            assert np.abs(got).max() < 1e-3 and np.abs(ref).max() < 1e-3, k
            continue
        err = nrm(got - ref) / (nrm(ref) + 1e-30)
        if err > worst:
            worst, worst_k = err, k
        assert err < GTOL["fp16"], "%s: relative L2 error %.3e (|ref| %.3e)" % (k, err, nrm(ref))
    print("\n[stage 1, bench shape %d / %d, 1 + %d titles, joint passes] total loss err %.2e  score err %.2e  worst gradient relative L2 error %.3e (%s)" % (
        Lt, Lb, Kn, le, se, worst, worst_k))
    assert le <= TOL["fp16"] and se <= TOL["fp16"]
