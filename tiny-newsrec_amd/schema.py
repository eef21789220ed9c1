"""state_dict key schema of the reference's model_bert.Model and the two model sizes used by the
golden cases, the benchmark and the smoke test (shared by bench.py, __graft_entry__.py and tests/)."""

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
