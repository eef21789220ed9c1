"""CPU: the numpy oracle against vectors captured from the imported reference (tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_case
from oracle import data_oracle as DO
from oracle import newsrec_oracle as O

RTOL, ATOL = 2e-4, 2e-5      # fp32 restatement vs fp32 reference (different summation orders)


def test_relpos_bucket_every_distance():
    z = np.load(os.path.join(GOLDEN, "relpos.npz"))
    assert (O.relative_position_bucket(z["rel"]) == z["bucket"]).all()
    for L in (24, 30, 128):
        assert np.array_equal(O.relpos_bias_table(z["weight"], L), z["table%d" % L])


def test_amsgrad_three_steps():
    z = np.load(os.path.join(GOLDEN, "amsgrad.npz"))
    for i in range(2):
        p = z["p0_%d" % i].copy()
        m, v, vm = np.zeros_like(p), np.zeros_like(p), np.zeros_like(p)
        for s in range(3):
            O.amsgrad_step(p, z["g%d_%d" % (s, i)], m, v, vm, s + 1, lr=1e-2)
            np.testing.assert_allclose(p, z["p%d_%d" % (s + 1, i)], rtol=1e-5, atol=1e-6)


def test_datapath_bit_exact():
    z = np.load(os.path.join(GOLDEN, "datapath.npz"))
    news_index = {str(k): int(v) for k, v in zip(z["news_ids"], z["news_index"])}
    comb = z["news_combined"]
    assert comb.dtype == np.int32 and (comb[0] == 0).all()
    lines = [str(x) for x in z["lines"]]
    h, m, c, y = DO.decode_batch(lines, news_index, 50, 4, labels=z["labels_seed7"])
    temb = None
    import hashinit
    temb = [hashinit.hash_normal(11, "dp_temb%d" % i, (comb.shape[0], 8)) for i in range(2)]
    out = DO.gather_batch(h, m, c, y, comb, temb)
    assert np.array_equal(out[0], z["log_ids"]) and out[0].dtype == np.int64
    assert np.array_equal(out[1], z["log_mask"])
    assert np.array_equal(out[2], z["input_ids"])
    assert np.array_equal(out[3], z["targets"])
    assert np.array_equal(out[4][0], z["th0"]) and np.array_equal(out[5][1], z["tc1"])
    # label draw reproduces random.randint order of dataloader.py:136
    import random
    random.seed(7)
    _, _, c2, y2 = DO.decode_batch(lines, news_index, 50, 4)
    assert np.array_equal(y2, z["targets"]) and np.array_equal(c2, c)
    d = os.path.join(GOLDEN, "data")
    for w in (1, 2, 3):
        for r in range(w):
            for sh in (0, 1):
                got = [os.path.basename(x) for x in DO.get_worker_files(d, r, w, "behaviors_np4_*.tsv", bool(sh), 3)]
                assert got == [str(x) for x in z["files_w%d_r%d_s%d" % (w, r, sh)]]


def test_datapath_known_answers():
    # captured from the reference
    ni = {"N10": 1, "N11": 2, "N12": 3, "N13": 4}
    line = "1\tU1\tt\tN10 N11 N999\tN12\tN13 N10 N11 N13"
    h, m, c, y = DO.decode_batch([line], ni, 50, 4, labels=[3])
    assert h[0].tolist() == [0] * 47 + [1, 2, 0] and m[0].tolist() == [0] * 47 + [1, 1, 1]
    assert c[0].tolist() == [4, 1, 2, 3, 4]
    x, mk = DO.pad_to_fix_len(list(range(60)), 50)
    assert x == list(range(10, 60)) and mk == [1] * 50
    assert DO.pad_to_fix_len([], 3) == ([0, 0, 0], [0, 0, 0])


def _check_model(name, full_grads):
    z, P, cfg, inp = load_case(name)
    out = O.model_fwd(P, cfg, *inp)
    for k, g in (("total_loss", "total"), ("distill_loss", "distill"), ("emb_loss", "emb"), ("target_loss", "target")):
        np.testing.assert_allclose(out[k], z[g], rtol=RTOL, atol=ATOL, err_msg=k)
    np.testing.assert_allclose(out["student_score"], z["score"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out["hist"], z["hist_vec"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out["cand"], z["cand_vec"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out["user"], z["user_vec"], rtol=RTOL, atol=ATOL)
    hid = out["cache"]["nc"]["hidden"]
    for li, h in enumerate(hid):
        ref = z["hidden%d" % li]
        np.testing.assert_allclose(h[:ref.shape[0]], ref, rtol=1e-3, atol=2e-4, err_msg="hidden%d" % li)
    G = O.model_bwd(P, cfg, out)
    names = [str(n) for n in z["grad_names"]]
    assert set(G.keys()) >= set(names), set(names) - set(G.keys())
    ref_of = lambda n: z["grad." + n] if full_grads else z["gval." + n]
    # Mathematical no-ops: a key bias adds a per-query constant that the softmax / the NRMS
    # normaliser removes, the fc2 bias of an additive pooling cancels in its normaliser.  The reference's gradient for
    # them is pure fp32 rounding noise, which makes them a measurement of the autograd noise floor of their block
    # (large for the NRMS user encoder, whose true gradients are tiny next to its activations).
    noop = lambda n: n.endswith("self.key.bias") or n.endswith("att_fc2.bias") or n.endswith("W_K.bias")
    block = lambda n: "user" if n.startswith("student.user_encoder.") else "news"
    floor = {"user": 0.0, "news": 0.0}
    for n in names:
        if noop(n):
            floor[block(n)] = max(floor[block(n)], float(np.abs(ref_of(n)).max()))
    for n in names:
        g = G[n]
        ref_norm = float(z["gnorm." + n])
        nf = 4.0 * floor[block(n)]
        if noop(n):
            assert np.sqrt((g.astype(np.float64) ** 2).sum()) < 1e-4 + nf * np.sqrt(g.size) and ref_norm < 1e-4
            continue
        np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), ref_norm, rtol=1e-3,
                                   atol=1e-9 + nf * np.sqrt(g.size), err_msg=n)
        scale = ref_norm / np.sqrt(g.size) + 1e-12
        scale = max(scale, float(np.abs(ref_of(n)).max()))
        got = g if full_grads else g.reshape(-1)[z["gidx." + n]]
        np.testing.assert_allclose(got, ref_of(n), rtol=2e-3, atol=2e-3 * scale + 1e-10 + nf, err_msg=n)
    # parameters outside the trainable set receive no gradient in the reference either
    lo = "encoder.layer.%d." % min(cfg["trainable_layers"])
    assert not any("embeddings" in k or "rel_pos" in k for k in G)


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "tiny_model_*.npz"))))
def test_tiny_model_forward_backward(name):
    _check_model(name, True)


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "full_model_*.npz"))))
def test_full_model_forward_backward(name):
    _check_model(name, False)


@pytest.mark.parametrize("name", ["stage1_tiny.npz", "stage1_full.npz", "stage1_cfg4.npz"])
def test_stage1_distill_forward_backward(name):
    """Stage-1 KD (title/body matching, bodies of 40 / 128 tokens) against the notebook's DistillModel."""
    from helpers import load_stage1_case
    z, P, cfg, inp = load_stage1_case(name)
    out = O.distill_fwd(P, cfg, *inp)
    for k, g in (("total_loss", "total"), ("target_loss", "target"), ("distill_loss", "distill"), ("emb_loss", "emb")):
        np.testing.assert_allclose(out[k], z[g], rtol=RTOL, atol=ATOL, err_msg=k)
    np.testing.assert_allclose(out["student_score"], z["score"], rtol=1e-3, atol=ATOL)
    G = O.distill_bwd(P, cfg, out)
    for n in [str(x) for x in z["grad_names"]]:
        g = G[n]
        ref_norm = float(z["gnorm." + n])
        if n.endswith("self.key.bias") or n.endswith("att_fc2.bias"):
            assert np.sqrt((g.astype(np.float64) ** 2).sum()) < 1e-4 and ref_norm < 1e-4
            continue
        np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), ref_norm, rtol=1e-3, atol=1e-9, err_msg=n)
        if "grad." + n in z.files:
            ref = z["grad." + n]
            np.testing.assert_allclose(g, ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()) + 1e-10, err_msg=n)
        else:
            ref = z["gval." + n]
            np.testing.assert_allclose(g.reshape(-1)[z["gidx." + n]], ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()) + 1e-10, err_msg=n)


def stage1_dropouts(z):
    """The two encoder passes' mask sets of a *_drop golden: titles = pass 0, bodies = pass 1, first forward call."""
    from oracle import dropout_oracle as DO
    p_h, p_a, seed = [float(x) for x in z["dropout"]]
    return DO.Dropout(p_h, p_a, int(seed), 0), DO.Dropout(p_h, p_a, int(seed), 1)


@pytest.mark.parametrize("name", ["stage1_tiny_drop.npz", "stage1_cfg4_drop.npz"])
def test_stage1_distill_train_mode_dropout(name):
    """Stage 1 as the notebook trains it -- .train(), dropout live (Post-train_KD.ipynb cell 19:6) -- against the notebook's own
    modules run with their nn.Dropout forwards replaced by the counter-based masks of oracle/dropout_oracle.py: pins WHERE the
    four dropout sites act and how they scale, forward and backward."""
    from helpers import load_stage1_case
    z, P, cfg, inp = load_stage1_case(name)
    dt, db = stage1_dropouts(z)
    out = O.distill_fwd(P, cfg, *inp, drop_title=dt, drop_body=db)
    for k, g in (("total_loss", "total"), ("target_loss", "target"), ("distill_loss", "distill"), ("emb_loss", "emb")):
        np.testing.assert_allclose(out[k], z[g], rtol=RTOL, atol=ATOL, err_msg=k)
    np.testing.assert_allclose(out["student_score"], z["score"], rtol=1e-3, atol=ATOL)
    # the masks matter: the eval-mode forward of the same inputs is somewhere else
    plain = O.distill_fwd(P, cfg, *inp, keep=False)
    assert np.abs(plain["student_score"] - z["score"]).max() > 20 * ATOL
    G = O.distill_bwd(P, cfg, out)
    for n in [str(x) for x in z["grad_names"]]:
        g = G[n]
        ref_norm = float(z["gnorm." + n])
        if n.endswith("self.key.bias") or n.endswith("att_fc2.bias"):
            assert np.sqrt((g.astype(np.float64) ** 2).sum()) < 1e-4 and ref_norm < 1e-4
            continue
        np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), ref_norm, rtol=1e-3, atol=1e-9, err_msg=n)
        if "grad." + n in z.files:
            ref = z["grad." + n]
            np.testing.assert_allclose(g, ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()) + 1e-10, err_msg=n)
        else:
            ref = z["gval." + n]
            np.testing.assert_allclose(g.reshape(-1)[z["gidx." + n]], ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()) + 1e-10, err_msg=n)


def test_philox_known_answers_and_mask_statistics():
    """Philox4x32-10 against the Random123 known-answer vectors, and the keep rate / scaling of the masks built on it."""
    from oracle import dropout_oracle as DO
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in DO.philox4x32_10(*ctr, *key)) == want
    m = DO.rows_mask(0.1, 99, DO.site_id(DO.KIND_FFN_OUT, 3), 5, 2048, 768)
    assert set(np.unique(m)) == {np.float32(0.0), np.float32(1.0 / 0.9)}
    assert abs((m == 0).mean() - DO.threshold(0.1) / 65536.0) < 1e-3 and abs(m.mean() - 1.0) < 2e-3
    pm = DO.probs_mask(0.1, 99, DO.site_id(DO.KIND_PROB, 1), 5, 8, 12, 30)
    assert abs((pm == 0).mean() - 0.1) < 3e-3
    # different sites / calls / seeds are different streams
    for other in (DO.rows_mask(0.1, 99, DO.site_id(DO.KIND_FFN_OUT, 2), 5, 2048, 768), DO.rows_mask(0.1, 99, DO.site_id(DO.KIND_FFN_OUT, 3), 6, 2048, 768),
                  DO.rows_mask(0.1, 98, DO.site_id(DO.KIND_FFN_OUT, 3), 5, 2048, 768)):
        assert 0.15 < ((m == 0) != (other == 0)).mean() < 0.21          # 2 p (1 - p) = 0.18 for independent masks


@pytest.mark.parametrize("model_type", ["bert", "roberta"])
def test_plmnr_bert_and_roberta_encoders(model_type):
    """PLM-NR --model_type bert / roberta (PLM-NR/utils.py:17-21, model_bert.py:109-118): transformers BertModel / RobertaModel
    as the news encoder = the oracle's encoder with a zero rel-pos table, RoBERTa with its position rule (cumulative non-pad
    count + padding_idx), one token type and layer_norm_eps 1e-5; loss, scores and every encoder-layer / head gradient."""
    from helpers import load_plmnr_hf_case
    z, P, cfg, inp = load_plmnr_hf_case(model_type)
    loss, score, out = O.plmnr_fwd(P, cfg, *inp, keep=True)
    np.testing.assert_allclose(float(loss), float(z["loss0"]), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(score, z["score0"], rtol=1e-3, atol=ATOL)
    G = O.plmnr_bwd(P, cfg, out)
    for n in [str(x) for x in z["grad_names"]]:
        g = G["student." + n.replace(".bert_model.", ".bert_model.bert.")]
        ref_norm = float(z["gnorm." + n])
        if n.endswith("self.key.bias") or n.endswith("att_fc2.bias"):
            assert np.sqrt((g.astype(np.float64) ** 2).sum()) < 1e-4 and ref_norm < 1e-4
            continue
        np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), ref_norm, rtol=1e-3, atol=1e-9, err_msg=n)
        ref = z["gval." + n]
        np.testing.assert_allclose(g.reshape(-1)[z["gidx." + n]], ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()) + 1e-10, err_msg=n)


def test_nrms_self_attention_backward_matches_finite_differences():
    """The NRMS goldens carry little gradient through the user encoder (see the noise-floor note above), so the
    hand-derived backward of model_bert.py:37-100 is also checked against central differences of its own forward."""
    rs = np.random.RandomState(0)
    B, U, D, nh = 2, 5, 32, 2
    x = rs.randn(B, U, D).astype(np.float32) * 0.5
    W = {k: (rs.randn(nh * 16, D) * 0.3).astype(np.float32) for k in "qkv"}
    b = {k: (rs.randn(nh * 16) * 0.1).astype(np.float32) for k in "qkv"}
    mask = np.array([[1, 1, 0, 1, 1], [0, 1, 1, 1, 0]], np.float32)
    R = rs.randn(B, U, nh * 16).astype(np.float32)
    for m in (None, mask):
        f = lambda x_, W_, b_: float((O.mhsa_fwd(x_, W_["q"], b_["q"], W_["k"], b_["k"], W_["v"], b_["v"], nh, m)[0].astype(np.float64) * R).sum())
        out, c = O.mhsa_fwd(x, W["q"], b["q"], W["k"], b["k"], W["v"], b["v"], nh, m)
        dx, G = O.mhsa_bwd(R, c, W["q"], W["k"], W["v"])
        eps = 2e-2

        def fd(arr, idx, rebuild):
            old = arr[idx]
            arr[idx] = old + eps
            hi = rebuild()
            arr[idx] = old - eps
            lo = rebuild()
            arr[idx] = old
            return (hi - lo) / (2 * eps)
        for idx in [(0, 1, 3), (1, 4, 31), (1, 2, 7)]:
            np.testing.assert_allclose(dx[idx], fd(x, idx, lambda: f(x, W, b)), rtol=3e-2, atol=3e-3)
        for key, name in (("q", "W_Q"), ("k", "W_K"), ("v", "W_V")):
            for idx in [(0, 0), (17, 5), (31, 31)]:
                np.testing.assert_allclose(G[name + ".weight"][idx], fd(W[key], idx, lambda: f(x, W, b)), rtol=3e-2, atol=3e-3)
            np.testing.assert_allclose(G[name + ".bias"][3], fd(b[key], 3, lambda: f(x, W, b)), rtol=3e-2, atol=3e-3)


@pytest.mark.parametrize("case", ["plmnr_full_0.npz", "plmnr_full_1.npz"])      # 2 layers ; configs[1]: 12 layers, train 10-11
def test_plmnr_training_steps_match_reference(case):
    """BASELINE configs[0]/[1]: PLM-NR ModelBert (CE only) forward, backward and two AMSGrad steps with the two
    learning rates of PLM-NR/run.py:104-106, against the reference's own run (tests/golden/plmnr_full_*.npz)."""
    from helpers import load_plmnr_case
    z, P, cfg, inp = load_plmnr_case(case)
    lr_bert, lr = [float(x) for x in z["lrs"]]
    state = {}
    for step in range(2):
        loss, score, out = O.plmnr_fwd(P, cfg, *inp, keep=True)
        np.testing.assert_allclose(loss, z["loss%d" % step], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(score, z["score%d" % step], rtol=2e-4, atol=2e-4)
        G = O.plmnr_bwd(P, cfg, out)
        if step == 0:
            names = [str(n) for n in z["grad_names"]]
            assert set("student." + n for n in names) == set(G)
            # 12-layer hash model: the q / k gradients are ~1e-8 against ~1e-2 for the FFN ones, i.e. at the fp32 rounding
            # floor of the reference's own autograd -- samples are held to 1e-5 of the model's largest gradient sample
            top = max(float(np.abs(z["gval." + n]).max()) for n in names)
            top_norm = max(float(z["gnorm." + n]) for n in names)
            checked = 0
            for n in names:
                g = G["student." + n]
                if n.endswith("self.key.bias") or n.endswith("att_fc2.bias"):
                    continue
                noisy = cfg["n_layers"] > 2 and float(z["gnorm." + n]) < 1e-4 * top_norm
                if not noisy:       # (as in the 12-layer stage-0 case below: norms of noise-level gradients are not a measurement)
                    np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), float(z["gnorm." + n]), rtol=1e-3,
                                               atol=1e-9, err_msg=n)
                    checked += 1
                ref = z["gval." + n]
                np.testing.assert_allclose(g.reshape(-1)[z["gidx." + n]], ref, rtol=2e-3,
                                           atol=2e-3 * np.abs(ref).max() + 1e-10 + (1e-5 * top if cfg["n_layers"] > 2 else 0.0), err_msg=n)
            assert checked >= 24          # at least every FFN / value / output / LayerNorm gradient of the two trainable layers
        for k, g in G.items():
            m, v, vm = state.setdefault(k, [np.zeros_like(P[k]), np.zeros_like(P[k]), np.zeros_like(P[k])])
            O.amsgrad_step(P[k], g, m, v, vm, step + 1, lr=lr_bert if ".bert_model." in k else lr)
    for k in [f[5:] for f in z.files if f.startswith("widx.")]:
        got = P["student." + k].reshape(-1)[z["widx." + k]]
        # Adam normalises the update to ~lr per element, so parameter samples pin the optimiser semantics tightly
        np.testing.assert_allclose(got, z["wval." + k], rtol=0, atol=2e-6, err_msg=k)


def test_training_trajectory_first_steps_match_reference():
    """trajectory_0.npz (the reference's own loop, Tiny-NewsRec/run.py:173-200, 50 steps over 5 fixed batches): the oracle's
    forward + backward + AMSGrad reproduce the first four steps' losses and scores (the GPU test replays all 50 in fp16)."""
    from helpers import load_trajectory_case
    z, P, cfg, batches, lr, steps = load_trajectory_case()
    assert steps == 50 and len(batches) == 5 and z["losses"].shape == (50, 4)
    assert z["losses"][-1, 0] < 0.6 * z["losses"][0, 0]                 # the reference did learn something over the 50 steps
    state = {}
    for step in range(4):
        out = O.model_fwd(P, cfg, *batches[step % len(batches)])
        got = [out["total_loss"], out["distill_loss"], out["emb_loss"], out["target_loss"]]
        np.testing.assert_allclose(np.array(got, np.float64), z["losses"][step], rtol=2e-4, atol=2e-5, err_msg="step %d" % step)
        np.testing.assert_allclose(out["student_score"], z["scores"][step], rtol=2e-4, atol=2e-4)
        G = O.model_bwd(P, cfg, out)
        for k, g in G.items():
            m, v, vm = state.setdefault(k, [np.zeros_like(P[k]), np.zeros_like(P[k]), np.zeros_like(P[k])])
            O.amsgrad_step(P[k], g, m, v, vm, step + 1, lr=lr)


def test_stage0_contrastive_post_train_matches_notebook():
    """Domian-specific_Post-train.ipynb TitleBodySimModel (12 layers, layers 9-11 trainable, CE over 1+K titles).
    With hash weights twelve layers deep the attention (and the pooling) is close to uniform, so the q / k and pooling-head
    gradients are ~1e-6 of the others and sit at the fp32 noise of the reference's own autograd (its mathematically-zero
    key-bias gradient has the same size): errors are judged against the largest gradient entry of the step."""
    from helpers import load_stage0_case
    z, P, cfg, inp = load_stage0_case()
    out = O.distill_fwd(P, cfg, *inp)
    np.testing.assert_allclose(out["total_loss"], z["total"], rtol=RTOL, atol=ATOL)
    assert float(out["distill_loss"]) == 0.0 and float(out["emb_loss"]) == 0.0
    np.testing.assert_allclose(out["student_score"], z["score"], rtol=1e-3, atol=ATOL)
    G = O.distill_bwd(P, cfg, out)
    names = [str(n) for n in z["grad_names"]]
    assert set("student." + n for n in names) == set(G)
    top = max(float(np.abs(z["gval." + n]).max()) for n in names)
    top_norm = max(float(z["gnorm." + n]) for n in names)
    checked = 0
    for n in names:
        g = G["student." + n]
        ref = z["gval." + n]
        np.testing.assert_allclose(g.reshape(-1)[z["gidx." + n]], ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()) + 1e-5 * top, err_msg=n)
        if float(z["gnorm." + n]) > 1e-4 * top_norm:
            np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), float(z["gnorm." + n]), rtol=1e-3, err_msg=n)
            checked += 1
    assert checked >= 36          # every FFN / value / output / LayerNorm / dense gradient of the three trainable layers


# ---------------------------------------------------------------------------------------------------------------
# oracle/torch_port.py: the torch-CPU port that bench.py times as `cpu_baseline` -- pinned to the same reference goldens
@pytest.mark.parametrize("name", ["full_model_0.npz", "full_model_1.npz"])
def test_torch_port_matches_reference_golden(name):
    import torch
    from oracle import torch_port as TP
    z, P, cfg, inp = load_case(name)
    tr = TP.Trainer(P, cfg, lr=1e-4)
    total, distill, emb, target, score = tr.step(*inp)
    for got, key in ((total, "total"), (distill, "distill"), (emb, "emb"), (target, "target")):
        np.testing.assert_allclose(float(got.detach()), z[key], rtol=RTOL, atol=ATOL, err_msg=key)
    np.testing.assert_allclose(score.detach().numpy(), z["score"], rtol=RTOL, atol=ATOL)
    names = [str(n) for n in z["grad_names"]]
    assert set(names) == {k for k, v in tr.P.items() if v.requires_grad}       # the trainable set of run.py:101-112
    for n in names:
        if n.endswith("self.key.bias") or n.endswith("att_fc2.bias"):
            continue                                                            # mathematical no-ops: rounding noise only
        g = tr.P[n].grad.numpy()
        np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), float(z["gnorm." + n]), rtol=1e-3, atol=1e-9, err_msg=n)


def test_torch_port_plmnr_two_rate_steps():
    """PLM-NR objective + the two learning rates of PLM-NR/run.py:104-106: parameters after two updates."""
    from helpers import load_plmnr_case
    from oracle import torch_port as TP
    z, P, cfg, inp = load_plmnr_case()
    lr_bert, lr = [float(x) for x in z["lrs"]]
    tr = TP.Trainer(P, cfg, lr=lr, lr_bert=lr_bert)
    for step in range(2):
        total, _, _, target, score = tr.step(*inp)
        np.testing.assert_allclose(float(target), z["loss%d" % step], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(score.detach().numpy(), z["score%d" % step], rtol=2e-4, atol=2e-4)
    for k in [f[5:] for f in z.files if f.startswith("widx.")]:
        got = tr.P["student." + k].detach().numpy().reshape(-1)[z["widx." + k]]
        np.testing.assert_allclose(got, z["wval." + k], rtol=0, atol=2e-6, err_msg=k)


def test_quality_golden_pins_the_oracle_eval_path_and_the_metrics():
    """tests/golden/quality_0.npz (reference trained 400 steps + evaluated by its own test(), make_golden.golden_quality): on the
    TRAINED weights the oracle's news encoder reproduces the reference's news_scoring (a 96-news sample: pad row, every topic), its
    eval user encoder the reference's user vectors, the product's metrics.py the reference's metrics.py (sklearn AUC included) on
    the reference's own scores - exactly - and the index-level decode the reference loader's label draws."""
    import random
    import sys
    import hashinit
    import metrics as PM
    from helpers import FULL, state_shapes
    sys.path.insert(0, GOLDEN)
    from quality_corpus import dequantize_delta, quality_corpus
    z = np.load(os.path.join(GOLDEN, "quality_0.npz"))
    seed, B, T_, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
    assert z["metrics"][0] > 0.75 and abs(z["metrics"] - z["metrics_unquantised"]).max() < 1e-3 and z["delta_quantisation_rel_l2"][0] < 0.01
    P = hashinit.init_state_dict(seed, state_shapes(FULL, nl, D, T_))
    for n in [str(x) for x in z["param_names"]]:
        P[n] = P[n] + dequantize_delta(z["dq." + n], z["ds." + n])
    comb = z["news_combined"].astype(np.int64)
    # the committed corpus is the generator's (the fixture travels, the generator documents it)
    c = quality_corpus(seed, T=T_, n_train=int(z["steps"][0]) * B)
    assert (c["news_combined"] == comb).all() and list(z["test_lines"]) == c["test_lines"] and list(z["train_lines"]) == c["train_lines"]
    for i in range(T_):
        assert (c["tables"][i] == z["table%d" % i]).all()
    rows = np.r_[0:8, 1 + 50 * np.arange(12), 300:376]
    got, _ = O.news_encoder_fwd(P, comb[rows], nl, A)
    np.testing.assert_allclose(got, z["news_scoring"][rows], rtol=2e-4, atol=2e-5)
    news_index = {"N%d" % i: i for i in range(1, comb.shape[0])}
    off, per = z["score_offsets"], z["per_impression"]
    sums, cnt = np.zeros(4), 0
    for i, ln in enumerate(z["test_lines"]):
        f = str(ln).split("\t")
        y = np.array([int(x.split("-")[1]) for x in f[4].split()])
        sc = z["scores"][off[i]:off[i + 1]]
        if i < 40:                                   # eval user encoder + scorer on the reference's news vectors
            h, m = DO.pad_to_fix_len(DO.trans_to_nindex(news_index, f[3].split()), U)
            cidx = DO.trans_to_nindex(news_index, [x.split("-")[0] for x in f[4].split()])
            u, _ = O.user_encoder_fwd(P, "student.user_encoder.", z["news_scoring"][np.array(h)][None], np.array(m, np.float32)[None], True)
            np.testing.assert_allclose(u[0], z["user_vecs"][i], rtol=2e-4, atol=2e-5)
            np.testing.assert_allclose(z["news_scoring"][np.array(cidx)] @ u[0], sc, rtol=2e-4, atol=2e-5)
        if y.mean() in (0, 1):
            assert np.isnan(per[i]).all()
            continue
        mine = [PM.roc_auc_score(y, sc), PM.mrr_score(y, sc), PM.ndcg_score(y, sc, 5), PM.ndcg_score(y, sc, 10)]
        np.testing.assert_allclose(mine, per[i], rtol=0, atol=1e-12)
        sums += mine
        cnt += 1
    np.testing.assert_allclose(sums / cnt, z["metrics"], rtol=0, atol=1e-12)
    assert cnt == 591
    # the label draws of the reference's loader (random.seed(seed), one randint per line): the oracle's and the product's decode
    from decode_worker import decode_lines
    lines = [str(l).encode() for l in z["train_lines"]]
    random.seed(seed)
    for step in range(int(z["steps"][0])):
        h, m, cd, y = DO.decode_batch(lines[step * B:(step + 1) * B], news_index, U, C - 1)
        assert (y == z["labels"][step]).all(), step
    random.seed(seed)
    h2, m2, c2, y2 = decode_lines(lines[:B], news_index, U, C - 1)
    h1, m1, c1, y1 = DO.decode_batch(lines[:B], news_index, U, C - 1, labels=z["labels"][0])
    assert (h1 == h2).all() and (m1 == m2).all() and (c1 == c2).all() and (y2 == z["labels"][0]).all()
    # and the training side at step 0: the oracle's four losses on the first batch against the reference's first logged step
    P0 = hashinit.init_state_dict(seed, state_shapes(FULL, nl, D, T_))
    tabs = [z["table%d" % i] for i in range(T_)]
    cfg = dict(n_layers=nl, heads=A, trainable_layers=[int(x) for x in z["trainable"]], user_log_mask=True, temperature=1.0, coef=0.2,
               pooling="att", nrms_heads=0)
    out = O.model_fwd(P0, cfg, comb[h1], m1, comb[c1], y1, [t[h1] for t in tabs], [t[c1] for t in tabs], keep=False)
    want = z["losses"][0]                              # total, distill, emb, target
    got = [out["total_loss"], out["distill_loss"], out["emb_loss"], out["target_loss"]]
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5)


def test_quality_long_continues_quality_0():
    """tests/golden/quality_long.npz (the reference's loop for 1 600 steps, its own test() every 200; tests/test_quality_gpu.py (iii))
    is the SAME run as quality_0.npz for its first 400 steps: same corpus lines, same label draws, the same 400 x 4 losses bit for
    bit, and its step-400 metrics are quality_0's unquantised ones to the run-to-run spread of the reference's CPU backward
    (make_golden.golden_quality: 1e-4)."""
    a, b = np.load(os.path.join(GOLDEN, "quality_0.npz")), np.load(os.path.join(GOLDEN, "quality_long.npz"))
    n = int(a["steps"][0])
    assert int(b["steps"][0]) == 4 * n and np.array_equal(a["meta"], b["meta"]) and float(a["lr"][0]) == float(b["lr"][0])
    assert np.array_equal(a["train_lines"], b["train_lines"][:len(a["train_lines"])]) and np.array_equal(a["test_lines"], b["test_lines"])
    assert np.array_equal(a["labels"], b["labels"][:n]) and np.array_equal(a["losses"], b["losses"][:n])
    i = list(b["metrics_at_steps"]).index(n)
    assert np.abs(b["metrics_at"][i] - a["metrics_unquantised"]).max() <= 1e-4
    # a plateau: from step 1 000 on the reference's AUC stays above 0.87 and within 1.5 pt (10 % of the dev labels are noise)
    auc = np.concatenate([b["metrics_at"][:, 0], b["metrics"][:1]])[-4:]
    assert auc.min() > 0.87 and auc.max() - auc.min() < 0.015
