"""ORACLE -- test infrastructure only, never a product path.

CPU (numpy, fp32) restatement of the Tiny-NewsRec data-parallel training hot
path: UniLMv2 news encoder -> additive-attention pooling -> user encoder ->
dot-product scorer -> stage-2 multi-teacher KD loss, with a hand-derived
backward and the AMSGrad update.  It is the checker the HIP path is compared
with (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) and nothing
under tiny-newsrec_amd/ may import it.

Pinning: the reference ships no tests or golden vectors, so
this file is pinned against outputs of the reference's own Python, imported
once in the build container by tests/golden/make_golden.py and committed as
tests/golden/*.npz (tests/test_oracle_golden.py, fp32, rtol 2e-4 / atol 2e-5).

Citations are file:line under /root/reference/Tiny-NewsRec/ unless prefixed.
Third-party arithmetic restated from transformers==3.0.2 (README.md:10):
BertSelfOutput / BertOutput = dense -> dropout(off) -> LayerNorm(x + residual),
BertIntermediate = dense -> erf-GELU (call sites tnlrv3/modeling.py:279,296-297).
"""
import math

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32
PFX = "student.news_encoder."
BERT = PFX + "bert_model.bert."


# --------------------------------------------------------------------------- #
# relative position bias  (tnlrv3/modeling.py:345-373, used at :458-463)
# --------------------------------------------------------------------------- #
# Bucket edges of the log branch for num_buckets=32 (16 per direction),
# max_exact=8, max_distance=128.  The reference evaluates
# 8 + int(log(n/8)/log(16)*8) in fp32; the integer edges below are what that
# produces (checked for every n in [0, 511] against the imported reference,
# tests/golden/relpos.npz) and avoid re-deriving fp32 log rounding at n=16,32,64.
_LOG_EDGES = (12, 16, 23, 32, 46, 64, 91)


def relative_position_bucket(rel, num_buckets=32, max_distance=128):
    """rel = key_pos - query_pos (int array) -> bucket ids, bidirectional."""
    assert num_buckets == 32 and max_distance == 128, "edges are tabulated for the tnlrv3 config"
    rel = np.asarray(rel, dtype=np.int64)
    half = num_buckets // 2
    n = np.abs(rel)
    large = 8 + np.searchsorted(np.asarray(_LOG_EDGES), n, side="right")
    large = np.minimum(large, half - 1)
    return (rel > 0).astype(np.int64) * half + np.where(n < 8, n, large)


def relpos_bias_table(weight, L):
    """weight (A, 32) = bert.rel_pos_bias.weight -> (A, L, L) additive bias.

    Equals the per-call (N, A, L, L) tensor of tnlrv3/modeling.py:459-463
    (one_hot @ Linear, permuted) for position_ids = arange(L)."""
    pos = np.arange(L)
    bucket = relative_position_bucket(pos[None, :] - pos[:, None])   # [i, j] = j - i
    return np.ascontiguousarray(weight[:, bucket]).astype(F32)


# --------------------------------------------------------------------------- #
# primitives
# --------------------------------------------------------------------------- #
def layer_norm_fwd(x, g, b, eps):
    mu = x.mean(-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True, dtype=F32)
    rstd = (1.0 / np.sqrt(var + F32(eps))).astype(F32)
    xh = xc * rstd
    return (xh * g + b).astype(F32), (xh, rstd)


def layer_norm_bwd(dy, cache, g):
    xh, rstd = cache
    dg = (dy * xh).reshape(-1, xh.shape[-1]).sum(0)
    db = dy.reshape(-1, xh.shape[-1]).sum(0)
    dxh = dy * g
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xh * (dxh * xh).mean(-1, keepdims=True))
    return dx.astype(F32), dg.astype(F32), db.astype(F32)


def gelu(x):
    return (x * 0.5 * (1.0 + _erf(x / F32(math.sqrt(2.0))))).astype(F32)


def gelu_grad(x):
    cdf = 0.5 * (1.0 + _erf(x / F32(math.sqrt(2.0))))
    pdf = np.exp(-0.5 * x * x) * F32(1.0 / math.sqrt(2.0 * math.pi))
    return (cdf + x * pdf).astype(F32)


def softmax(x, axis=-1):
    m = x.max(axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis, keepdims=True)).astype(F32)


def log_softmax(x, axis=-1):
    m = x.max(axis, keepdims=True)
    z = x - m
    return (z - np.log(np.exp(z).sum(axis, keepdims=True))).astype(F32)


def linear(x, w, b=None):
    y = x @ w.T
    return y + b if b is not None else y


# --------------------------------------------------------------------------- #
# encoder  (tnlrv3/modeling.py:133-178, 181-342, 421-476)
# --------------------------------------------------------------------------- #
def embeddings_fwd(P, ids, eps=1e-12, drop=None, pos_pad=None):
    """BertEmbeddings.forward tnlrv3/modeling.py:153-178: word + pos[arange] + type[0] -> LN [-> dropout, train mode, :177].
    drop: oracle/dropout_oracle.Dropout (the masks of this forward call) or None."""
    L = ids.shape[1]
    if pos_pad is None:
        pe = P[BERT + "embeddings.position_embeddings.weight"][:L][None]
    else:       # RoBERTa (PLM-NR --model_type roberta): transformers create_position_ids_from_input_ids
        ne = (ids != pos_pad).astype(np.int64)
        pe = P[BERT + "embeddings.position_embeddings.weight"][np.cumsum(ne, 1) * ne + pos_pad]
    e = (P[BERT + "embeddings.word_embeddings.weight"][ids]
         + pe
         + P[BERT + "embeddings.token_type_embeddings.weight"][0][None, None])
    y, _ = layer_norm_fwd(e.astype(F32), P[BERT + "embeddings.LayerNorm.weight"],
                          P[BERT + "embeddings.LayerNorm.bias"], eps)
    if drop is not None:
        m = drop.hidden(0, 0, y.shape[0] * y.shape[1], y.shape[2])
        if m is not None:
            y = (y * m.reshape(y.shape)).astype(F32)
    return y


def _lp(l):
    return BERT + "encoder.layer.%d." % l


def bert_layer_fwd(P, l, x, mask_add, rel, A, eps=1e-12, drop=None):
    """BertLayer.forward tnlrv3/modeling.py:299-308 (+ BertSelfAttention :205-272).

    x (N,L,H); mask_add (N,L) = (1-mask)*-10000 (:454); rel (A,L,L).
    drop (train mode only): dropout on the attention probabilities (:224) and on the two output Linears before their residual
    adds (transformers BertSelfOutput / BertOutput, call sites :287, :306); masks from oracle/dropout_oracle.py."""
    p = _lp(l)
    N, L, H = x.shape
    d = H // A
    q = linear(x, P[p + "attention.self.query.weight"], P[p + "attention.self.query.bias"])
    k = linear(x, P[p + "attention.self.key.weight"], P[p + "attention.self.key.bias"])
    v = linear(x, P[p + "attention.self.value.weight"], P[p + "attention.self.value.bias"])
    qh = q.reshape(N, L, A, d).transpose(0, 2, 1, 3)
    kh = k.reshape(N, L, A, d).transpose(0, 2, 1, 3)
    vh = v.reshape(N, L, A, d).transpose(0, 2, 1, 3)
    s = (qh @ kh.transpose(0, 1, 3, 2)) / F32(math.sqrt(d))
    s = s + mask_add[:, None, None, :] + rel[None]
    pr = softmax(s, -1)
    mp = drop.probs(l, N, A, L) if drop is not None else None
    mo = drop.hidden(2, l, N * L, H) if drop is not None else None
    mf = drop.hidden(3, l, N * L, H) if drop is not None else None
    prd = (pr * mp).astype(F32) if mp is not None else pr           # what multiplies V
    ctx = (prd @ vh).transpose(0, 2, 1, 3).reshape(N, L, H)
    ao = linear(ctx, P[p + "attention.output.dense.weight"], P[p + "attention.output.dense.bias"])
    if mo is not None:
        ao = ao * mo.reshape(N, L, H)
    h1, ln1 = layer_norm_fwd((ao + x).astype(F32), P[p + "attention.output.LayerNorm.weight"],
                             P[p + "attention.output.LayerNorm.bias"], eps)
    u = linear(h1, P[p + "intermediate.dense.weight"], P[p + "intermediate.dense.bias"]).astype(F32)
    g = gelu(u)
    f = linear(g, P[p + "output.dense.weight"], P[p + "output.dense.bias"])
    if mf is not None:
        f = f * mf.reshape(N, L, H)
    y, ln2 = layer_norm_fwd((f + h1).astype(F32), P[p + "output.LayerNorm.weight"],
                            P[p + "output.LayerNorm.bias"], eps)
    cache = dict(x=x, qh=qh, kh=kh, vh=vh, pr=pr, ctx=ctx, ln1=ln1, h1=h1, u=u, g=g, ln2=ln2, mp=mp, mo=mo, mf=mf)
    return y, cache


def bert_layer_bwd(P, l, dy, c, A, need_dx=True, need_dw=True):
    """Backward of bert_layer_fwd.  Returns (dx or None, {param: grad})."""
    p = _lp(l)
    G = {}
    x = c["x"]
    N, L, H = x.shape
    d = H // A
    M = N * L
    r2 = lambda t: t.reshape(M, -1)
    dypre, dg2, db2 = layer_norm_bwd(dy, c["ln2"], P[p + "output.LayerNorm.weight"])
    dres2 = dypre                                                  # gradient of the residual branch (h1)
    if c.get("mf") is not None:                                    # gradient of the Linear's output: through its dropout
        dypre = (dypre * c["mf"].reshape(dypre.shape)).astype(F32)
    dgact = dypre @ P[p + "output.dense.weight"]
    du = (dgact * gelu_grad(c["u"])).astype(F32)
    dh1 = dres2 + du @ P[p + "intermediate.dense.weight"]
    dh1pre, dg1, db1 = layer_norm_bwd(dh1.astype(F32), c["ln1"], P[p + "attention.output.LayerNorm.weight"])
    dres1 = dh1pre                                                 # residual branch (x)
    if c.get("mo") is not None:
        dh1pre = (dh1pre * c["mo"].reshape(dh1pre.shape)).astype(F32)
    dctx = dh1pre @ P[p + "attention.output.dense.weight"]
    dch = dctx.reshape(N, L, A, d).transpose(0, 2, 1, 3)
    pr, qh, kh, vh = c["pr"], c["qh"], c["kh"], c["vh"]
    dp = dch @ vh.transpose(0, 1, 3, 2)
    if c.get("mp") is not None:
        dp = dp * c["mp"]
        dvh = (pr * c["mp"]).transpose(0, 1, 3, 2) @ dch
    else:
        dvh = pr.transpose(0, 1, 3, 2) @ dch
    ds = pr * (dp - (dp * pr).sum(-1, keepdims=True))
    sc = F32(1.0 / math.sqrt(d))
    dqh = (ds @ kh) * sc
    dkh = (ds.transpose(0, 1, 3, 2) @ qh) * sc
    back = lambda t: t.transpose(0, 2, 1, 3).reshape(N, L, H)
    dq, dk, dv = back(dqh), back(dkh), back(dvh)
    if need_dw:
        G[p + "output.LayerNorm.weight"], G[p + "output.LayerNorm.bias"] = dg2, db2
        G[p + "output.dense.weight"] = r2(dypre).T @ r2(c["g"])
        G[p + "output.dense.bias"] = r2(dypre).sum(0)
        G[p + "intermediate.dense.weight"] = r2(du).T @ r2(c["h1"])
        G[p + "intermediate.dense.bias"] = r2(du).sum(0)
        G[p + "attention.output.LayerNorm.weight"], G[p + "attention.output.LayerNorm.bias"] = dg1, db1
        G[p + "attention.output.dense.weight"] = r2(dh1pre).T @ r2(c["ctx"])
        G[p + "attention.output.dense.bias"] = r2(dh1pre).sum(0)
        for nm, t in (("query", dq), ("key", dk), ("value", dv)):
            G[p + "attention.self.%s.weight" % nm] = r2(t).T @ r2(x)
            G[p + "attention.self.%s.bias" % nm] = r2(t).sum(0)
        G = {k_: v_.astype(F32) for k_, v_ in G.items()}
    dx = None
    if need_dx:
        dx = (dres1 + dq @ P[p + "attention.self.query.weight"] + dk @ P[p + "attention.self.key.weight"]
              + dv @ P[p + "attention.self.value.weight"]).astype(F32)
    return dx, G


# --------------------------------------------------------------------------- #
# AttentionPooling  (model_bert.py:8-34)
# --------------------------------------------------------------------------- #
def att_pool_fwd(x, w1, b1, w2, b2, mask=None):
    """alpha = exp(fc2(tanh(fc1 x))) [*mask] / (sum + 1e-8); out = sum alpha x.  No max-subtraction."""
    e = np.tanh(linear(x, w1, b1)).astype(F32)                 # (n,k,Q)
    al = np.exp(linear(e, w2, b2))[..., 0].astype(F32)         # (n,k)
    if mask is not None:
        al = al * mask
    den = al.sum(1, keepdims=True) + F32(1e-8)
    w = (al / den).astype(F32)
    out = (w[..., None] * x).sum(1).astype(F32)
    return out, dict(x=x, e=e, al=al, den=den, w=w, mask=mask)


def att_pool_bwd(dout, c, w1, w2):
    x, e, w, den = c["x"], c["e"], c["w"], c["den"]
    dx = w[..., None] * dout[:, None, :]
    dw = (x * dout[:, None, :]).sum(-1)                        # (n,k)
    dal = (dw - (dw * w).sum(1, keepdims=True)) / den
    da = dal * c["al"]          # al already carries the mask factor: d/da (exp(a) m) = exp(a) m
    de = da[..., None] * w2[0][None, None, :]
    dpre = (de * (1.0 - e * e)).astype(F32)
    Q = e.shape[-1]
    g_w2 = (da[..., None] * e).reshape(-1, Q).sum(0)[None, :]
    g_b2 = da.sum().reshape(1)
    g_w1 = dpre.reshape(-1, Q).T @ x.reshape(-1, x.shape[-1])
    g_b1 = dpre.reshape(-1, Q).sum(0)
    dx = dx + dpre @ w1
    return dx.astype(F32), g_w1.astype(F32), g_b1.astype(F32), g_w2.astype(F32), g_b2.astype(F32)


# --------------------------------------------------------------------------- #
# NewsEncoder / UserEncoder / ModelBert  (model_bert.py:103-205)
# --------------------------------------------------------------------------- #
def split_tokens(x2l):
    """(N, 2L) -> ids (N,L), mask (N,L)   model_bert.py:124-127."""
    L = x2l.shape[1] // 2
    return x2l[:, :L], x2l[:, L:]


def encoder_fwd(P, ids, mask, n_layers, A, keep_from=None, drop=None, eps=1e-12, pos_pad=None):
    """TuringNLRv3Model.forward tnlrv3/modeling.py:421-476 -> last hidden state.

    keep_from: first layer whose cache is kept for backward (None = keep none)."""
    mask_add = ((1.0 - mask.astype(F32)) * F32(-10000.0)).astype(F32)
    x = embeddings_fwd(P, ids, eps=eps, drop=drop, pos_pad=pos_pad)
    rel = relpos_bias_table(P[BERT + "rel_pos_bias.weight"], ids.shape[1])
    caches = {}
    hidden = [x]
    for l in range(n_layers):
        x, c = bert_layer_fwd(P, l, x, mask_add, rel, A, eps=eps, drop=drop)
        hidden.append(x)
        if keep_from is not None and l >= keep_from:
            caches[l] = c
    return x, caches, hidden


def news_encoder_fwd(P, x2l, n_layers, A, keep_from=None, pooling="att", drop=None, eps=1e-12, pos_pad=None):
    """NewsEncoder.forward model_bert.py:119-137: pooling 'att' (additive attention, no mask) | 'cls' (token 0) |
    anything else = mean over ALL L positions (padding included, :135), then dense."""
    ids, mask = split_tokens(x2l)
    h, caches, hidden = encoder_fwd(P, ids, mask, n_layers, A, keep_from, drop=drop, eps=eps, pos_pad=pos_pad)
    pc = None
    if pooling == "att":
        nv, pc = att_pool_fwd(h, P[PFX + "attn.att_fc1.weight"], P[PFX + "attn.att_fc1.bias"],
                              P[PFX + "attn.att_fc2.weight"], P[PFX + "attn.att_fc2.bias"])
    elif pooling == "cls":
        nv = h[:, 0, :].astype(F32)
    else:
        nv = h.mean(1, dtype=F32)
    out = linear(nv, P[PFX + "dense.weight"], P[PFX + "dense.bias"]).astype(F32)
    return out, dict(layers=caches, pool=pc, nv=nv, hidden=hidden, pooling=pooling, L=h.shape[1])


def news_encoder_bwd(P, dout, c, A, trainable_layers):
    G = {}
    G[PFX + "dense.weight"] = (dout.T @ c["nv"]).astype(F32)
    G[PFX + "dense.bias"] = dout.sum(0).astype(F32)
    dnv = dout @ P[PFX + "dense.weight"]
    pooling, L = c.get("pooling", "att"), c.get("L")
    if pooling == "att":
        dh, g1, gb1, g2, gb2 = att_pool_bwd(dnv, c["pool"], P[PFX + "attn.att_fc1.weight"],
                                            P[PFX + "attn.att_fc2.weight"])
        G[PFX + "attn.att_fc1.weight"], G[PFX + "attn.att_fc1.bias"] = g1, gb1
        G[PFX + "attn.att_fc2.weight"], G[PFX + "attn.att_fc2.bias"] = g2, gb2
    elif pooling == "cls":
        dh = np.zeros((dnv.shape[0], L, dnv.shape[1]), F32)
        dh[:, 0, :] = dnv
    else:
        dh = np.repeat((dnv / F32(L))[:, None, :], L, 1).astype(F32)
    if trainable_layers:
        lo = min(trainable_layers)
        for l in sorted(c["layers"].keys(), reverse=True):
            dh, g = bert_layer_bwd(P, l, dh, c["layers"][l], A, need_dx=(l > lo),
                                   need_dw=(l in trainable_layers))
            G.update(g)
    return G


# --------------------------------------------------------------------------- #
# NRMS user encoder: MultiHeadSelfAttention over the clicked-news vectors (model_bert.py:37-100), d_k = d_v = 16
# --------------------------------------------------------------------------- #
NRMS_DK = 16


def mhsa_fwd(x, wq, bq, wk, bk, wv, bv, n_heads, mask=None):
    """x (B,U,D) -> (B,U,n_heads*16).  scores = exp(QK^T / 4) [* mask over keys] / (sum + 1e-8): raw exp, no
    max-subtraction (model_bert.py:51-58)."""
    B, U, _ = x.shape
    dk = NRMS_DK
    sp = lambda t: t.reshape(B, U, n_heads, dk).transpose(0, 2, 1, 3)
    q, k, v = sp(linear(x, wq, bq)), sp(linear(x, wk, bk)), sp(linear(x, wv, bv))
    sc = np.exp(np.einsum("bhid,bhjd->bhij", q, k).astype(F32) / F32(np.sqrt(dk))).astype(F32)
    if mask is not None:
        sc = sc * mask[:, None, None, :].astype(F32)
    den = sc.sum(-1, keepdims=True) + F32(1e-8)
    attn = (sc / den).astype(F32)
    ctx = np.einsum("bhij,bhjd->bhid", attn, v).astype(F32)
    out = ctx.transpose(0, 2, 1, 3).reshape(B, U, n_heads * dk)
    return out.astype(F32), dict(x=x, q=q, k=k, v=v, attn=attn, den=den, sc=sc, H=n_heads)


def mhsa_bwd(dout, c, wq, wk, wv):
    """-> dx, {W_Q,W_K,W_V}.{weight,bias} gradients."""
    x, q, k, v, attn, den, nh = c["x"], c["q"], c["k"], c["v"], c["attn"], c["den"], c["H"]
    B, U, D = x.shape
    dk = NRMS_DK
    dctx = dout.reshape(B, U, nh, dk).transpose(0, 2, 1, 3)
    dattn = np.einsum("bhid,bhjd->bhij", dctx, v)
    dv = np.einsum("bhij,bhid->bhjd", attn, dctx)
    # attn = sc / den, den = sum_j sc + eps  ->  dsc = (dattn - sum_j dattn * attn) / den ; the mask factor is part of sc
    dsc = (dattn - (dattn * attn).sum(-1, keepdims=True)) / den
    ds = dsc * c["sc"] / F32(np.sqrt(dk))            # d exp(s) = exp(s) (already masked) ds
    dq = np.einsum("bhij,bhjd->bhid", ds, k)
    dkk = np.einsum("bhij,bhid->bhjd", ds, q)
    mg = lambda t: t.transpose(0, 2, 1, 3).reshape(B * U, nh * dk).astype(F32)
    dq2, dk2, dv2 = mg(dq), mg(dkk), mg(dv)
    x2 = x.reshape(B * U, D)
    G = {"W_Q.weight": dq2.T @ x2, "W_Q.bias": dq2.sum(0), "W_K.weight": dk2.T @ x2, "W_K.bias": dk2.sum(0),
         "W_V.weight": dv2.T @ x2, "W_V.bias": dv2.sum(0)}
    dx = (dq2 @ wq + dk2 @ wk + dv2 @ wv).reshape(B, U, D)
    return dx.astype(F32), {n: g.astype(F32) for n, g in G.items()}


def user_encoder_fwd(P, pfx, news_vecs, log_mask, user_log_mask, nrms_heads=0):
    """UserEncoder.forward model_bert.py:155-176.  nrms_heads > 0: args.model == 'NRMS' (self-attention over the
    clicked news before the additive pooling, :162-163 / :171-172)."""
    w1, b1 = P[pfx + "attn.att_fc1.weight"], P[pfx + "attn.att_fc1.bias"]
    w2, b2 = P[pfx + "attn.att_fc2.weight"], P[pfx + "attn.att_fc2.bias"]
    mh = pfx + "multi_head_self_attn."
    sa = lambda x, m: mhsa_fwd(x, P[mh + "W_Q.weight"], P[mh + "W_Q.bias"], P[mh + "W_K.weight"], P[mh + "W_K.bias"],
                               P[mh + "W_V.weight"], P[mh + "W_V.bias"], nrms_heads, m)
    if user_log_mask:
        x, sc = (sa(news_vecs.astype(F32), log_mask) if nrms_heads else (news_vecs, None))
        out, c = att_pool_fwd(x, w1, b1, w2, b2, mask=log_mask)
        c["blend"], c["sa"] = False, sc
        return out, c
    m = log_mask[..., None]
    hv = (news_vecs * m + P[pfx + "pad_doc"][None] * (1.0 - m)).astype(F32)
    x, sc = (sa(hv, None) if nrms_heads else (hv, None))
    out, c = att_pool_fwd(x, w1, b1, w2, b2)
    c["blend"], c["sa"] = True, sc
    c["m"] = m
    return out, c


def user_encoder_bwd(P, pfx, dout, c):
    dhv, g1, gb1, g2, gb2 = att_pool_bwd(dout, c, P[pfx + "attn.att_fc1.weight"], P[pfx + "attn.att_fc2.weight"])
    G = {pfx + "attn.att_fc1.weight": g1, pfx + "attn.att_fc1.bias": gb1,
         pfx + "attn.att_fc2.weight": g2, pfx + "attn.att_fc2.bias": gb2}
    if c.get("sa") is not None:
        mh = pfx + "multi_head_self_attn."
        dhv, gs = mhsa_bwd(dhv, c["sa"], P[mh + "W_Q.weight"], P[mh + "W_K.weight"], P[mh + "W_V.weight"])
        for n, g in gs.items():
            G[mh + n] = g
    if c["blend"]:
        m = c["m"]
        G[pfx + "pad_doc"] = (dhv * (1.0 - m)).sum((0, 1))[None].astype(F32)
        dnews = (dhv * m).astype(F32)
    else:
        G[pfx + "pad_doc"] = np.zeros_like(P[pfx + "pad_doc"])
        dnews = dhv
    return dnews, G


def cross_entropy_rows(score, label):
    """F.cross_entropy(reduction='none')."""
    ls = log_softmax(score, -1)
    return -ls[np.arange(score.shape[0]), label]


# --------------------------------------------------------------------------- #
# Model.forward: stage-2 multi-teacher KD  (model_bert.py:262-305)
# --------------------------------------------------------------------------- #
def model_fwd(P, cfg, history, history_mask, candidate, label, teacher_hist, teacher_cand, keep=True):
    """Returns dict with total/distill/emb/target losses, student_score and a cache.

    cfg: dict(n_layers, heads, trainable_layers, user_log_mask, temperature, coef[, pooling='att'|'cls'|'mean',
    nrms_heads=0 (args.model == 'NRMS': args.num_attention_heads)]).
    history (B,U,2L) int; history_mask (B,U) f32; candidate (B,C,2L); label (B,);
    teacher_hist / teacher_cand: lists of (B,U,D) / (B,C,D)."""
    B, U, W2 = history.shape
    C = candidate.shape[1]
    A, nl = cfg["heads"], cfg["n_layers"]
    tr = sorted(cfg["trainable_layers"])
    keep_from = (min(tr) if tr else nl) if keep else None
    # ModelBert.forward :187-205 -- candidates and history share the encoder; rows are independent,
    # so one pass over the concatenation equals the reference's two calls.
    allx = np.concatenate([history.reshape(B * U, W2), candidate.reshape(B * C, W2)], 0)
    vec, nc = news_encoder_fwd(P, allx, nl, A, keep_from, cfg.get("pooling", "att"), eps=cfg.get("ln_eps", 1e-12),
                               pos_pad=cfg.get("pos_pad"))
    nrms = int(cfg.get("nrms_heads", 0))
    D = vec.shape[1]
    hist = vec[:B * U].reshape(B, U, D)
    cand = vec[B * U:].reshape(B, C, D)
    user, uc = user_encoder_fwd(P, "student.user_encoder.", hist, history_mask, cfg["user_log_mask"], nrms)
    score = np.einsum("bcd,bd->bc", cand, user).astype(F32)
    S = np.concatenate([hist, cand], 1)                              # :270
    target = cross_entropy_rows(score, label).mean(dtype=F32)         # :271
    T = len(teacher_hist)
    t_scores, t_losses, NE, UE, projs, tus, tups, Tcat = [], [], [], [], [], [], [], []
    for i in range(T):
        tn = np.concatenate([teacher_hist[i], teacher_cand[i]], 1).astype(F32)      # :277
        W, b = P["transform_matrix.%d.weight" % i], P["transform_matrix.%d.bias" % i]
        pr = linear(tn, W, b).astype(F32)                                           # :278
        NE.append(((S - pr) ** 2).mean(-1).mean(-1))                                # :279-280
        tu, _ = user_encoder_fwd(P, "teachers.%d." % i, teacher_hist[i].astype(F32), history_mask,
                                 cfg["user_log_mask"], nrms)                        # :282
        tup = linear(tu, W, b).astype(F32)                                          # :283
        UE.append(((user - tup) ** 2).mean(-1))                                     # :284
        ts = np.einsum("bcd,bd->bc", teacher_cand[i].astype(F32), tu).astype(F32)   # :286-287
        t_scores.append(ts)
        t_losses.append(cross_entropy_rows(ts, label))                              # :288
        projs.append(pr); tus.append(tu); tups.append(tup); Tcat.append(tn)
    tau = F32(cfg["temperature"])
    if T:
        tw = softmax(-np.stack(t_losses, -1), -1)                                       # :292-293
        ts_mix = np.einsum("bct,bt->bc", np.stack(t_scores, -1), tw).astype(F32)        # :295-297
        pT = softmax(ts_mix / tau, -1)
        distill = (-(pT * log_softmax(score / tau, -1)).sum(-1)).mean(dtype=F32)        # :208-219
        NEs, UEs = np.stack(NE, -1), np.stack(UE, -1)
        emb = (NEs * tw).sum(-1).mean(dtype=F32) + (UEs * tw).sum(-1).mean(dtype=F32)   # :300-303
    else:       # no teachers: the PLM-NR objective (PLM-NR/model_bert.py:206), total = coef * CE
        tw, ts_mix, pT = np.zeros((B, 0), F32), np.zeros_like(score), None
        distill = emb = F32(0.0)
    total = distill + F32(cfg["coef"]) * target + emb                               # :305
    out = dict(total_loss=F32(total), distill_loss=F32(distill), emb_loss=F32(emb), target_loss=F32(target),
               student_score=score, hist=hist, cand=cand, user=user, teacher_weights=tw,
               teacher_scores=ts_mix)
    out["cache"] = dict(nc=nc, uc=uc, S=S, projs=projs, tus=tus, tups=tups, Tcat=Tcat, pT=pT, tw=tw,
                        label=label, B=B, U=U, C=C, D=D)
    return out


def model_bwd(P, cfg, out):
    """d total_loss / d every trainable parameter -> {state_dict key: grad}.

    Trainable set = run.py:101-112: heads + transform_matrix + encoder.layer[i], i in trainable_layers.
    teacher_weights do not depend on any trainable parameter (teachers frozen, :101-102)."""
    c = out["cache"]
    B, U, C, D = c["B"], c["U"], c["C"], c["D"]
    tw, S, label = c["tw"], c["S"], c["label"]
    score, user, cand = out["student_score"], out["user"], out["cand"]
    tau, coef = F32(cfg["temperature"]), F32(cfg["coef"])
    G = {}
    onehot = np.zeros_like(score)
    onehot[np.arange(B), label] = 1.0
    dscore = coef * (softmax(score, -1) - onehot) / F32(B)
    if c["pT"] is not None:
        dscore = dscore + (softmax(score / tau, -1) - c["pT"]) / tau / F32(B)
    dS = np.zeros_like(S)
    duser = np.zeros_like(user)
    npos = S.shape[1]
    for i in range(len(c["projs"])):
        w_i = tw[:, i]
        dne = (2.0 / (npos * D * B)) * w_i[:, None, None] * (S - c["projs"][i])      # d emb / d S
        dS += dne
        due = (2.0 / (D * B)) * w_i[:, None] * (user - c["tups"][i])
        duser += due
        gW = (-dne).reshape(-1, D).T @ c["Tcat"][i].reshape(-1, D) + (-due).T @ c["tus"][i]
        gb = (-dne).reshape(-1, D).sum(0) + (-due).sum(0)
        G["transform_matrix.%d.weight" % i] = gW.astype(F32)
        G["transform_matrix.%d.bias" % i] = gb.astype(F32)
    dcand = dS[:, U:].copy() + dscore[:, :, None] * user[:, None, :]
    duser = duser + np.einsum("bc,bcd->bd", dscore, cand)
    dhist_u, gu = user_encoder_bwd(P, "student.user_encoder.", duser.astype(F32), c["uc"])
    G.update(gu)
    dhist = dS[:, :U] + dhist_u
    dvec = np.concatenate([dhist.reshape(B * U, D), dcand.reshape(B * C, D)], 0).astype(F32)
    G.update(news_encoder_bwd(P, dvec, c["nc"], cfg["heads"], set(cfg["trainable_layers"])))
    return G


# --------------------------------------------------------------------------- #
# Stage-1 KD: DistillModel.forward  (Post-train_KD.ipynb cell 14:13-44, TitleBodySimModel cell 12)
# As published the cell multiplies a Python list by a tensor (cell 14:41); restated with the evident intent
# torch.stack(teacher_MSEs, -1), the form model_bert.py:300 uses.
# --------------------------------------------------------------------------- #
def distill_fwd(P, cfg, title, body, label, teacher_titles, teacher_bodies, keep=True, drop_title=None, drop_body=None):
    """title (B,1+K,2Lt) int, body (B,2Lb) int, label (B,), teacher_titles T x (B,1+K,D), teacher_bodies T x (B,D).
    cfg: n_layers, heads, trainable_layers.  loss = target + distill(tau=1) + emb.
    drop_title / drop_body: dropout masks of the two encoder passes (the notebook trains under .train(), cell 19:6)."""
    B, C, W2 = title.shape
    A, nl = cfg["heads"], cfg["n_layers"]
    tr = sorted(cfg["trainable_layers"])
    keep_from = (min(tr) if tr else nl) if keep else None
    bvec, bc = news_encoder_fwd(P, body, nl, A, keep_from, drop=drop_body)                     # cell 12: body first
    tvec, tc = news_encoder_fwd(P, title.reshape(B * C, W2), nl, A, keep_from, drop=drop_title)
    D = bvec.shape[1]
    tv = tvec.reshape(B, C, D)
    score = np.einsum("bcd,bd->bc", tv, bvec).astype(F32)
    target = cross_entropy_rows(score, label).mean(dtype=F32)
    T = len(teacher_titles)
    t_scores, t_losses, mses, ptit, pbod = [], [], [], [], []
    for i in range(T):
        tt, tb = teacher_titles[i].astype(F32), teacher_bodies[i].astype(F32)
        ts = np.einsum("bcd,bd->bc", tt, tb).astype(F32)
        t_scores.append(ts)
        t_losses.append(cross_entropy_rows(ts, label))
        W, b = P["transform_matrix.%d.weight" % i], P["transform_matrix.%d.bias" % i]
        pt, pb = linear(tt, W, b).astype(F32), linear(tb, W, b).astype(F32)
        mses.append(((tv - pt) ** 2).mean(-1).mean(-1) + ((bvec - pb) ** 2).mean(-1))
        ptit.append(pt); pbod.append(pb)
    if T:
        tw = softmax(-np.stack(t_losses, -1), -1)
        mix = np.einsum("bct,bt->bc", np.stack(t_scores, -1), tw).astype(F32)
        pT = softmax(mix, -1)
        distill = (-(pT * log_softmax(score, -1)).sum(-1)).mean(dtype=F32)
        emb = (np.stack(mses, -1) * tw).sum(-1).mean(dtype=F32)
    else:       # no teachers: stage 0, TitleBodySimModel of Domian-specific_Post-train.ipynb cell 11 (plain CE)
        tw, pT = np.zeros((B, 0), F32), None
        distill = emb = F32(0.0)
    out = dict(total_loss=F32(target + distill + emb), target_loss=F32(target), distill_loss=F32(distill), emb_loss=F32(emb),
               student_score=score, title_vec=tv, body_vec=bvec, teacher_weights=tw)
    out["cache"] = dict(bc=bc, tc=tc, pT=pT, tw=tw, ptit=ptit, pbod=pbod, label=label, B=B, C=C, D=D,
                        tt=[x.astype(F32) for x in teacher_titles], tb=[x.astype(F32) for x in teacher_bodies])
    return out


def distill_bwd(P, cfg, out):
    c = out["cache"]
    B, C, D, tw, label = c["B"], c["C"], c["D"], c["tw"], c["label"]
    score, tv, bvec = out["student_score"], out["title_vec"], out["body_vec"]
    onehot = np.zeros_like(score)
    onehot[np.arange(B), label] = 1.0
    dscore = (softmax(score, -1) - onehot) / F32(B)
    if c["pT"] is not None:
        dscore = dscore + (softmax(score, -1) - c["pT"]) / F32(B)
    dtv = dscore[:, :, None] * bvec[:, None, :]
    dbv = np.einsum("bc,bcd->bd", dscore, tv)
    G = {}
    for i in range(len(c["ptit"])):
        w_i = tw[:, i]
        dt = (2.0 / (C * D * B)) * w_i[:, None, None] * (tv - c["ptit"][i])
        db_ = (2.0 / (D * B)) * w_i[:, None] * (bvec - c["pbod"][i])
        dtv = dtv + dt
        dbv = dbv + db_
        G["transform_matrix.%d.weight" % i] = ((-dt).reshape(-1, D).T @ c["tt"][i].reshape(-1, D) + (-db_).T @ c["tb"][i]).astype(F32)
        G["transform_matrix.%d.bias" % i] = ((-dt).reshape(-1, D).sum(0) + (-db_).sum(0)).astype(F32)
    trl = set(cfg["trainable_layers"])
    g1 = news_encoder_bwd(P, dtv.reshape(B * C, D).astype(F32), c["tc"], cfg["heads"], trl)
    g2 = news_encoder_bwd(P, dbv.astype(F32), c["bc"], cfg["heads"], trl)
    for k in g1:
        G[k] = (g1[k] + g2[k]).astype(F32)
    return G


# --------------------------------------------------------------------------- #
# PLM-NR ModelBert.forward  (PLM-NR/model_bert.py:187-207): same encoders + CE
# --------------------------------------------------------------------------- #
def plmnr_fwd(P, cfg, history, history_mask, candidate, label, keep=False):
    """-> (loss, score[, full output incl. cache when keep]).  P uses the Tiny-NewsRec key names (student. prefix):
    PLM-NR's ModelBert is the same module tree without it."""
    z = model_fwd(P, dict(cfg, temperature=1.0, coef=1.0), history, history_mask, candidate, label, [], [],
                  keep=keep)
    return (z["target_loss"], z["student_score"], z) if keep else (z["target_loss"], z["student_score"])


def plmnr_bwd(P, cfg, out):
    """Gradients of the PLM-NR loss w.r.t. its trainable parameters (PLM-NR/run.py:84-91 freeze policy = A17)."""
    return model_bwd(P, dict(cfg, temperature=1.0, coef=1.0), out)


# --------------------------------------------------------------------------- #
# optimiser  (run.py:134: torch.optim.Adam(lr, amsgrad=True), torch defaults)
# --------------------------------------------------------------------------- #
def amsgrad_step(p, g, m, v, vmax, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam(amsgrad=True) update, in place; `step` is the 1-based count after increment."""
    m *= F32(b1); m += F32(1 - b1) * g
    v *= F32(b2); v += F32(1 - b2) * g * g
    np.maximum(vmax, v, out=vmax)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = np.sqrt(vmax) / F32(math.sqrt(bc2)) + F32(eps)
    p -= F32(lr / bc1) * (m / denom)
    return p


# --------------------------------------------------------------------------- #
# utils.acc  (utils.py:79-83)
# --------------------------------------------------------------------------- #
def acc(y_true, y_hat):
    return F32((np.argmax(y_hat, -1) == y_true).sum() / y_true.shape[0])
