"""Shared by CPU and GPU tests: rebuild a golden case's weights / inputs / config from its npz."""
import os

import numpy as np

import hashinit

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY = dict(H=64, A=4, I=256, vocab=128, max_pos=64, Q=16)
FULL = dict(H=768, A=12, I=3072, vocab=30522, max_pos=512, Q=200)


def _nrms_shapes(s, pfx, D, heads):
    for n in ("W_Q", "W_K", "W_V"):
        s[pfx + "multi_head_self_attn.%s.weight" % n] = (heads * 16, D)
        s[pfx + "multi_head_self_attn.%s.bias" % n] = (heads * 16,)


def state_shapes(dims, n_layers, D, T, pooling="att", nrms_heads=0):
    """state_dict key schema of model_bert.Model; pooling != 'att' drops the news encoder's
    additive attention, nrms_heads > 0 (args.model == 'NRMS') adds the user encoders' self-attention."""
    H, I, Q = dims["H"], dims["I"], dims["Q"]
    Du = nrms_heads * 16 if nrms_heads else D
    s = {}
    for i in range(T):
        if nrms_heads:
            _nrms_shapes(s, "teachers.%d." % i, D, nrms_heads)
        s["teachers.%d.pad_doc" % i] = (1, D)
        s["teachers.%d.attn.att_fc1.weight" % i] = (Q, Du)
        s["teachers.%d.attn.att_fc1.bias" % i] = (Q,)
        s["teachers.%d.attn.att_fc2.weight" % i] = (1, Q)
        s["teachers.%d.attn.att_fc2.bias" % i] = (1,)
    b = "student.news_encoder.bert_model.bert."
    s[b + "embeddings.word_embeddings.weight"] = (dims["vocab"], H)
    s[b + "embeddings.position_embeddings.weight"] = (dims["max_pos"], H)
    s[b + "embeddings.token_type_embeddings.weight"] = (2, H)
    s[b + "embeddings.LayerNorm.weight"] = (H,)
    s[b + "embeddings.LayerNorm.bias"] = (H,)
    for l in range(n_layers):
        p = b + "encoder.layer.%d." % l
        for nm in ("query", "key", "value"):
            s[p + "attention.self.%s.weight" % nm] = (H, H)
            s[p + "attention.self.%s.bias" % nm] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    s[b + "pooler.dense.weight"] = (H, H)
    s[b + "pooler.dense.bias"] = (H,)
    s[b + "rel_pos_bias.weight"] = (dims["A"], 32)
    s["student.news_encoder.bert_model.classifier.weight"] = (2, H)
    s["student.news_encoder.bert_model.classifier.bias"] = (2,)
    if pooling == "att":
        s["student.news_encoder.attn.att_fc1.weight"] = (Q, H)
        s["student.news_encoder.attn.att_fc1.bias"] = (Q,)
        s["student.news_encoder.attn.att_fc2.weight"] = (1, Q)
        s["student.news_encoder.attn.att_fc2.bias"] = (1,)
    s["student.news_encoder.dense.weight"] = (D, H)
    s["student.news_encoder.dense.bias"] = (D,)
    if nrms_heads:
        _nrms_shapes(s, "student.user_encoder.", D, nrms_heads)
    s["student.user_encoder.pad_doc"] = (1, D)
    s["student.user_encoder.attn.att_fc1.weight"] = (Q, Du)
    s["student.user_encoder.attn.att_fc1.bias"] = (Q,)
    s["student.user_encoder.attn.att_fc2.weight"] = (1, Q)
    s["student.user_encoder.attn.att_fc2.bias"] = (1,)
    for i in range(T):
        s["transform_matrix.%d.weight" % i] = (D, D)
        s["transform_matrix.%d.bias" % i] = (D,)
    return s


def load_case(name):
    """-> (z npz, P weights, cfg, inputs tuple)."""
    z = np.load(os.path.join(GOLDEN, name))
    seed, B, T, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
    dims = TINY if name.startswith("tiny") else FULL
    pooling, model, nh = [str(x) for x in z["variant"]] if "variant" in z.files else ("att", "NAML", "0")
    P = hashinit.init_state_dict(seed, state_shapes(dims, nl, D, T, pooling, int(nh)))
    ulm, tau, coef = [float(x) for x in z["flags"]]
    cfg = dict(n_layers=nl, heads=A, trainable_layers=[int(x) for x in z["trainable"]],
               user_log_mask=bool(ulm), temperature=tau, coef=coef, pooling=pooling, nrms_heads=int(nh))
    inp = (z["in_hist"], z["in_mask"], z["in_cand"], z["in_label"],
           [z["in_th%d" % i] for i in range(T)], [z["in_tc%d" % i] for i in range(T)])
    return z, P, cfg, inp


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


def load_stage0_case(name="stage0_full.npz"):
    """Stage-0 golden (Domian-specific_Post-train.ipynb TitleBodySimModel, 12 layers, CE only) -> (z, P, cfg, inputs)."""
    z = np.load(os.path.join(GOLDEN, name))
    seed, B, T, C, Lt, Lb, D, A, nl = [int(x) for x in z["meta"]]
    shapes = {k: v for k, v in state_shapes(FULL, nl, D, 0).items() if k.startswith("student.news_encoder.")}
    P = hashinit.init_state_dict(seed, shapes)
    cfg = dict(n_layers=nl, heads=A, trainable_layers=[int(x) for x in z["trainable"]])
    return z, P, cfg, (z["in_title"], z["in_body"], z["in_label"], [], [])
