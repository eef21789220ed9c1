"""Shared by CPU and GPU tests: rebuild a golden case's weights / inputs / config from its npz."""
import os

import numpy as np

import hashinit

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

from schema import FULL, TINY, state_shapes  # noqa: E402,F401  (kept importable from tests.helpers)


def test_hooks():
    """libtnr_testhooks.so (csrc/testhooks.hip): the CU hog used by the contention tests.  Not part of the product library."""
    import ctypes
    import tnr_hip
    L = ctypes.CDLL(os.path.join(os.path.dirname(tnr_hip.LIB_PATH), "libtnr_testhooks.so"))
    L.tnr_debug_cu_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.tnr_debug_cu_hog.restype = ctypes.c_int
    return L


def load_case(name):
    """-> (z npz, P weights, cfg, inputs tuple)."""
    z = np.load(os.path.join(GOLDEN, name))
    seed, B, T, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
    dims = TINY if name.startswith("tiny") else FULL
    pooling, model, nh = [str(x) for x in z["variant"]] if "variant" in z.files else ("att", "NAML", "0")
    P = hashinit.init_state_dict(seed, state_shapes(dims, nl, D, T, pooling, int(nh)))
    if "stats" in z.files and str(z["stats"][0]) == "pretrained_like":
        hashinit.pretrained_like(P, seed)
    ulm, tau, coef = [float(x) for x in z["flags"]]
    cfg = dict(n_layers=nl, heads=A, trainable_layers=[int(x) for x in z["trainable"]],
               user_log_mask=bool(ulm), temperature=tau, coef=coef, pooling=pooling, nrms_heads=int(nh))
    inp = (z["in_hist"], z["in_mask"], z["in_cand"], z["in_label"],
           [z["in_th%d" % i] for i in range(T)], [z["in_tc%d" % i] for i in range(T)])
    return z, P, cfg, inp


def load_trajectory_case(name="trajectory_0.npz"):
    """Reference training trajectory (make_golden.golden_trajectory) -> (z, P, cfg, [batch inputs], lr, steps)."""
    z = np.load(os.path.join(GOLDEN, name))
    seed, B, T, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
    P = hashinit.init_state_dict(seed, state_shapes(FULL, nl, D, T))
    ulm, tau, coef = [float(x) for x in z["flags"]]
    cfg = dict(n_layers=nl, heads=A, trainable_layers=[int(x) for x in z["trainable"]], user_log_mask=bool(ulm), temperature=tau,
               coef=coef, pooling="att", nrms_heads=0)
    batches = [(z["in_hist_b%d" % i], z["in_mask_b%d" % i], z["in_cand_b%d" % i], z["in_label_b%d" % i],
                [z["in_th%d_b%d" % (j, i)] for j in range(T)], [z["in_tc%d_b%d" % (j, i)] for j in range(T)])
               for i in range(int(z["n_batches"][0]))]
    return z, P, cfg, batches, float(z["lr"][0]), int(z["steps"][0])


def load_stage1_case(name):
    """Stage-1 KD golden (Post-train_KD.ipynb DistillModel) -> (z, P, cfg, inputs)."""
    z = np.load(os.path.join(GOLDEN, name))
    seed, B, T, C, Lt, Lb, D, A, nl = [int(x) for x in z["meta"]]
    dims = dict(TINY, Q=16) if name.startswith("stage1_tiny") else FULL
    P = hashinit.init_state_dict(seed, state_shapes(dims, nl, D, T))
    cfg = dict(n_layers=nl, heads=A, trainable_layers=[int(x) for x in z["trainable"]])
    inp = (z["in_title"], z["in_body"], z["in_label"], [z["in_tt%d" % i] for i in range(T)], [z["in_tb%d" % i] for i in range(T)])
    return z, P, cfg, inp


def unilm_checkpoint(seed, H, n_layers, A, I, vocab, max_pos):
    """A checkpoint in the published unilm2 layout (fused qkv_linear, q_bias / v_bias, encoder.rel_pos_bias),
    values from the shared hash generator."""
    t = lambda name, shape: __import__('torch').from_numpy(hashinit.hash_normal(seed, name, shape, 0.02))
    sd = {"bert.embeddings.word_embeddings.weight": t("we", (vocab, H)),
          "bert.embeddings.position_embeddings.weight": t("pe", (max_pos, H)),
          "bert.embeddings.token_type_embeddings.weight": t("te", (2, H)),
          "bert.embeddings.LayerNorm.weight": t("elw", (H,)), "bert.embeddings.LayerNorm.bias": t("elb", (H,)),
          "bert.encoder.rel_pos_bias.weight": t("rp", (A, 32)),
          "bert.pooler.dense.weight": t("pw", (H, H)), "bert.pooler.dense.bias": t("pb", (H,)),
          "cls.predictions.bias": t("cls", (vocab,))}
    for l in range(n_layers):
        p = "bert.encoder.layer.%d." % l
        sd[p + "attention.self.qkv_linear.weight"] = t(p + "qkv", (3 * H, H))
        sd[p + "attention.self.q_bias"] = t(p + "qb", (1, 1, H))
        sd[p + "attention.self.v_bias"] = t(p + "vb", (1, 1, H))
        for n, shp in (("attention.output.dense.weight", (H, H)), ("attention.output.dense.bias", (H,)),
                       ("attention.output.LayerNorm.weight", (H,)), ("attention.output.LayerNorm.bias", (H,)),
                       ("intermediate.dense.weight", (I, H)), ("intermediate.dense.bias", (I,)),
                       ("output.dense.weight", (H, I)), ("output.dense.bias", (H,)),
                       ("output.LayerNorm.weight", (H,)), ("output.LayerNorm.bias", (H,))):
            sd[p + n] = t(p + n, shp)
    return sd


def load_plmnr_case(name="plmnr_full_0.npz"):
    """PLM-NR golden (ModelBert + CE, two-lr AMSGrad) -> (z, P with Tiny-NewsRec key names, cfg, inputs)."""
    z = np.load(os.path.join(GOLDEN, name))
    seed, B, T, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
    P = hashinit.init_state_dict(seed, state_shapes(FULL, nl, D, 0))
    cfg = dict(n_layers=nl, heads=A, trainable_layers=[int(x) for x in z["trainable"]], user_log_mask=False,
               temperature=1.0, coef=1.0)
    return z, P, cfg, (z["in_hist"], z["in_mask"], z["in_cand"], z["in_label"])


def load_plmnr_hf_case(model_type):
    """PLM-NR ModelBert with --model_type bert / roberta (transformers BertModel / RobertaModel as the encoder) ->
    (z, P under the engine's / oracle's internal key names, cfg, inputs).  The reference's keys sit directly under bert_model.*;
    internally the encoder keeps the UniLM layout with the rel-pos bias (and the unused classification head) at zero."""
    z = np.load(os.path.join(GOLDEN, "plmnr_%s.npz" % model_type))
    seed, B, T, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
    vocab, max_pos, type_vocab = [int(x) for x in z["dims"]]
    shapes = state_shapes(dict(FULL, vocab=vocab, max_pos=max_pos), nl, D, 0)
    shapes["student.news_encoder.bert_model.bert.embeddings.token_type_embeddings.weight"] = (type_vocab, FULL["H"])
    P = {}
    for k, shp in shapes.items():
        if k.endswith("rel_pos_bias.weight") or ".bert_model.classifier." in k:
            P[k] = np.zeros(shp, np.float32)
        else:
            P[k] = hashinit.init_tensor(seed, k.replace(".bert_model.bert.", ".bert_model."), tuple(shp))
    ref_keys = {"student." + str(k) for k in z["keys"]}
    mine = {k.replace(".bert_model.bert.", ".bert_model.") for k in shapes if not (k.endswith("rel_pos_bias.weight") or ".bert_model.classifier." in k)}
    assert ref_keys == mine, ref_keys ^ mine
    cfg = dict(n_layers=nl, heads=A, trainable_layers=[int(x) for x in z["trainable"]], user_log_mask=False, temperature=1.0, coef=1.0,
               ln_eps=float(z["ln_eps"]), pos_pad=1 if model_type == "roberta" else None,
               vocab=vocab, max_pos=max_pos, type_vocab=type_vocab)
    return z, P, cfg, (z["in_hist"], z["in_mask"], z["in_cand"], z["in_label"])


def load_stage0_case(name="stage0_full.npz"):
    """Stage-0 golden (Domian-specific_Post-train.ipynb TitleBodySimModel, 12 layers, CE only) -> (z, P, cfg, inputs)."""
    z = np.load(os.path.join(GOLDEN, name))
    seed, B, T, C, Lt, Lb, D, A, nl = [int(x) for x in z["meta"]]
    shapes = {k: v for k, v in state_shapes(FULL, nl, D, 0).items() if k.startswith("student.news_encoder.")}
    P = hashinit.init_state_dict(seed, shapes)
    cfg = dict(n_layers=nl, heads=A, trainable_layers=[int(x) for x in z["trainable"]])
    return z, P, cfg, (z["in_title"], z["in_body"], z["in_label"], [], [])
