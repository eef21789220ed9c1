"""ORACLE -- test infrastructure only, never a product path.

torch-CPU port of the reference training step (forward by the same formulas as newsrec_oracle.py, backward by
torch autograd, torch.optim.Adam(amsgrad=True) as run.py:134): the strongest CPU implementation available on the
GPU box, where /root/reference does not exist.  It is what bench.py's `cpu_baseline` leg times (kind "port") and is
pinned against the reference's golden vectors in tests/test_oracle_golden.py.  Nothing under tiny-newsrec_amd/ may
import it.

Citations are file:line under /root/reference/Tiny-NewsRec/ (same lines as newsrec_oracle.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import newsrec_oracle as O

PFX, BERT = O.PFX, O.BERT


def is_trainable(name, trainable_layers):
    """run.py:101-112: teachers frozen; bert_model frozen except encoder.layer[i], i in trainable_layers."""
    if name.startswith("teachers."):
        return False
    if name.startswith(PFX + "bert_model."):
        return any(name.startswith(BERT + "encoder.layer.%d." % l) for l in trainable_layers)
    return True


def make_params(P, trainable_layers):
    """numpy state_dict -> {key: torch tensor}, requires_grad on the trainable set."""
    out = {}
    for k, v in P.items():
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).clone()
        t.requires_grad_(is_trainable(k, trainable_layers))
        out[k] = t
    return out


def _layer(P, l, x, mask_add, rel, A, eps=1e-12):
    """BertLayer.forward tnlrv3/modeling.py:299-308 (+ :205-272); x (N,L,H)."""
    p = BERT + "encoder.layer.%d." % l
    N, L, H = x.shape
    d = H // A
    sp = lambda t: t.view(N, L, A, d).transpose(1, 2)
    q = sp(F.linear(x, P[p + "attention.self.query.weight"], P[p + "attention.self.query.bias"]))
    k = sp(F.linear(x, P[p + "attention.self.key.weight"], P[p + "attention.self.key.bias"]))
    v = sp(F.linear(x, P[p + "attention.self.value.weight"], P[p + "attention.self.value.bias"]))
    s = q @ k.transpose(-1, -2) / math.sqrt(d) + mask_add[:, None, None, :] + rel[None]
    ctx = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(N, L, H)
    ao = F.linear(ctx, P[p + "attention.output.dense.weight"], P[p + "attention.output.dense.bias"])
    h1 = F.layer_norm(ao + x, (H,), P[p + "attention.output.LayerNorm.weight"], P[p + "attention.output.LayerNorm.bias"], eps)
    g = F.gelu(F.linear(h1, P[p + "intermediate.dense.weight"], P[p + "intermediate.dense.bias"]))
    f = F.linear(g, P[p + "output.dense.weight"], P[p + "output.dense.bias"])
    return F.layer_norm(f + h1, (H,), P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], eps)


def news_encoder(P, x2l, n_layers, A, eps=1e-12):
    """NewsEncoder.forward model_bert.py:119-137 (attention pooling without mask, then dense)."""
    L = x2l.shape[1] // 2
    ids, mask = x2l[:, :L], x2l[:, L:]
    H = P[BERT + "embeddings.LayerNorm.weight"].shape[0]
    e = (P[BERT + "embeddings.word_embeddings.weight"][ids] + P[BERT + "embeddings.position_embeddings.weight"][:L][None]
         + P[BERT + "embeddings.token_type_embeddings.weight"][0][None, None])
    x = F.layer_norm(e, (H,), P[BERT + "embeddings.LayerNorm.weight"], P[BERT + "embeddings.LayerNorm.bias"], eps)
    mask_add = (1.0 - mask.float()) * -10000.0
    rel = torch.from_numpy(O.relpos_bias_table(P[BERT + "rel_pos_bias.weight"].detach().numpy(), L))
    for l in range(n_layers):
        x = _layer(P, l, x, mask_add, rel, A, eps)
    nv = att_pool(x, P[PFX + "attn.att_fc1.weight"], P[PFX + "attn.att_fc1.bias"], P[PFX + "attn.att_fc2.weight"],
                  P[PFX + "attn.att_fc2.bias"])
    return F.linear(nv, P[PFX + "dense.weight"], P[PFX + "dense.bias"])


def att_pool(x, w1, b1, w2, b2, mask=None):
    """AttentionPooling.forward model_bert.py:15-34: raw exp, no max-subtraction, + 1e-8."""
    al = torch.exp(F.linear(torch.tanh(F.linear(x, w1, b1)), w2, b2))[..., 0]
    if mask is not None:
        al = al * mask
    al = al / (al.sum(1, keepdim=True) + 1e-8)
    return (al[..., None] * x).sum(1)


def user_encoder(P, pfx, vecs, mask, user_log_mask):
    """UserEncoder.forward model_bert.py:155-176 (NAML branch)."""
    w = (P[pfx + "attn.att_fc1.weight"], P[pfx + "attn.att_fc1.bias"], P[pfx + "attn.att_fc2.weight"], P[pfx + "attn.att_fc2.bias"])
    if user_log_mask:
        return att_pool(vecs, *w, mask=mask)
    m = mask[..., None]
    return att_pool(vecs * m + P[pfx + "pad_doc"][None] * (1.0 - m), *w)


def model_forward(P, cfg, history, history_mask, candidate, label, teacher_hist, teacher_cand):
    """Model.forward model_bert.py:262-305 (T = 0: the PLM-NR objective, PLM-NR/model_bert.py:187-207).
    -> (total, distill, emb, target, score) as torch scalars / tensor."""
    B, U, W2 = history.shape
    C = candidate.shape[1]
    allx = torch.cat([history.reshape(B * U, W2), candidate.reshape(B * C, W2)], 0)
    vec = news_encoder(P, allx, cfg["n_layers"], cfg["heads"])
    D = vec.shape[1]
    hist, cand = vec[:B * U].view(B, U, D), vec[B * U:].view(B, C, D)
    user = user_encoder(P, "student.user_encoder.", hist, history_mask, cfg["user_log_mask"])
    score = torch.bmm(cand, user[:, :, None])[..., 0]
    S = torch.cat([hist, cand], 1)
    target = F.cross_entropy(score, label)
    T = len(teacher_hist)
    zero = torch.zeros(())
    if T == 0:
        return cfg["coef"] * target, zero, zero, target, score
    ts_all, tl, NE, UE = [], [], [], []
    for i in range(T):
        W, b = P["transform_matrix.%d.weight" % i], P["transform_matrix.%d.bias" % i]
        pr = F.linear(torch.cat([teacher_hist[i], teacher_cand[i]], 1), W, b)
        NE.append(((S - pr) ** 2).mean(-1).mean(-1))
        tu = user_encoder(P, "teachers.%d." % i, teacher_hist[i], history_mask, cfg["user_log_mask"])
        UE.append(((user - F.linear(tu, W, b)) ** 2).mean(-1))
        ts = torch.bmm(teacher_cand[i], tu[:, :, None])[..., 0]
        ts_all.append(ts)
        tl.append(F.cross_entropy(ts, label, reduction="none"))
    tw = torch.softmax(-torch.stack(tl, -1), -1)
    mix = (torch.stack(ts_all, -1) * tw[:, None, :]).sum(-1)
    tau = cfg["temperature"]
    distill = (-(torch.softmax(mix / tau, -1) * torch.log_softmax(score / tau, -1)).sum(-1)).mean()
    emb = (torch.stack(NE, -1) * tw).sum(-1).mean() + (torch.stack(UE, -1) * tw).sum(-1).mean()
    return distill + cfg["coef"] * target + emb, distill, emb, target, score


class Trainer:
    """The reference loop body (run.py:178-195) on torch CPU: forward -> backward -> Adam(amsgrad) step."""

    def __init__(self, P_numpy, cfg, lr=1e-4, lr_bert=None):
        self.cfg = cfg
        self.P = make_params(P_numpy, cfg["trainable_layers"])
        train = [(k, v) for k, v in self.P.items() if v.requires_grad]
        if lr_bert is None:
            groups = [{"params": [v for _, v in train]}]
        else:        # PLM-NR/run.py:104-106: two learning rates
            groups = [{"params": [v for k, v in train if ".bert_model." in k], "lr": lr_bert},
                      {"params": [v for k, v in train if ".bert_model." not in k], "lr": lr}]
        self.opt = torch.optim.Adam(groups, lr=lr, amsgrad=True)

    def step(self, history, history_mask, candidate, label, teacher_hist=(), teacher_cand=()):
        t = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x)).to(dt) if dt else torch.from_numpy(np.ascontiguousarray(x))
        out = model_forward(self.P, self.cfg, t(history, torch.int64), t(history_mask, torch.float32), t(candidate, torch.int64),
                            t(label, torch.int64), [t(x, torch.float32) for x in teacher_hist], [t(x, torch.float32) for x in teacher_cand])
        self.opt.zero_grad(set_to_none=True)
        out[0].backward()
        self.opt.step()
        return out
