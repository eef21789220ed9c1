"""Generate tests/golden/*.npz by executing the REFERENCE's own Python in this container.

Run once here (the reference is read-only at /root/reference and does not
exist on the GPU box):  python tests/golden/make_golden.py
Only inputs/outputs (data) are written; weights come from the shared
deterministic generator tiny-newsrec_amd/hashinit.py, so no weight files and no
reference text are committed.
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
sys.path.insert(0, ROOT)
import hashinit  # noqa: E402
import ref_shim  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

TINY_CFG = dict(ref_shim.BASE_CFG, hidden_size=64, num_attention_heads=4, intermediate_size=256,
                vocab_size=128, max_position_embeddings=64, num_hidden_layers=2)
TINY_ARGS = dict(news_dim=32, news_query_vector_dim=16, user_query_vector_dim=16, user_log_length=6,
                 npratio=2, num_words_title=10)


def fill(model, seed, stats=None):
    sd = model.state_dict()
    W = {k: hashinit.init_tensor(seed, k, tuple(v.shape)) for k, v in sd.items()}
    if stats == "pretrained_like":
        hashinit.pretrained_like(W, seed)
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(torch.from_numpy(W[k]))
    return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


def make_inputs(seed, B, U, C, L, vocab, T, D, n_news=40):
    """MIND-shaped synthetic batch: (history (B,U,2L), mask (B,U), candidate (B,C,2L), label, teacher lists)."""
    lens = hashinit.hash_randint(seed, "len", (n_news + 1,), 3, L + 1)
    toks = hashinit.hash_randint(seed, "tok", (n_news + 1, L), 1, vocab)
    comb = np.zeros((n_news + 1, 2 * L), np.int64)
    for i in range(1, n_news + 1):
        comb[i, :lens[i]] = toks[i, :lens[i]]
        comb[i, L:L + lens[i]] = 1
    hl = hashinit.hash_randint(seed, "hl", (B,), 0, U + 1)
    hl[0] = U
    if B > 1:
        hl[1] = 0                                  # empty history: every slot is the pad news
    hidx = hashinit.hash_randint(seed, "hidx", (B, U), 1, n_news + 1)
    mask = np.zeros((B, U), np.float32)
    for b in range(B):
        hidx[b, :U - hl[b]] = 0
        mask[b, U - hl[b]:] = 1
    if B > 2 and hl[2] > 0:
        hidx[2, U - 1] = 0                         # unknown id inside the history: index 0 but mask 1
    cidx = hashinit.hash_randint(seed, "cidx", (B, C), 1, n_news + 1)
    label = hashinit.hash_randint(seed, "label", (B,), 0, C)
    temb = [hashinit.hash_normal(seed, "temb%d" % i, (n_news + 1, D), std=0.3) for i in range(T)]
    return (comb[hidx], mask, comb[cidx], label, [t[hidx] for t in temb], [t[cidx] for t in temb])


def grad_samples(seed, name, g, k=64):
    flat = g.reshape(-1)
    idx = hashinit.hash_randint(seed, "gs." + name, (min(k, flat.size),), 0, flat.size)
    return idx, flat[idx]


def run_model(R, cfg_json, args_over, trainable, seed, B, T, full_grads, stats=None):
    a = ref_shim.make_args(config_name=ref_shim.write_config(cfg_json), num_teachers=T, batch_size=B,
                           **args_over)
    model = R.model_bert.Model(a)
    P = fill(model, seed, stats)
    # run.py:101-112 freeze policy
    for p in model.teachers.parameters():
        p.requires_grad = False
    for p in model.student.news_encoder.bert_model.parameters():
        p.requires_grad = False
    for i, layer in enumerate(model.student.news_encoder.bert_model.bert.encoder.layer):
        if i in trainable:
            for p in layer.parameters():
                p.requires_grad = True
    L = a.num_words_title
    inp = make_inputs(seed, B, a.user_log_length, a.npratio + 1, L, cfg_json["vocab_size"], T, a.news_dim)
    hist, mask, cand, label, th, tc = inp
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    out = model(tt(hist), tt(mask), tt(cand), tt(label), [tt(x) for x in th], [tt(x) for x in tc])
    total, distill, emb, target, score = out
    total.backward()
    # intermediate tensors for op-by-op pinning
    with torch.no_grad():
        s2, hv, cv, uv = model.student(tt(hist), tt(mask), tt(cand))
        ids = tt(np.concatenate([hist.reshape(-1, 2 * L), cand.reshape(-1, 2 * L)], 0))
        bo = model.student.news_encoder.bert_model(ids[:, :L], ids[:, L:])
        hidden = [h.numpy() for h in bo[3]]
    rec = dict(total=total.item(), distill=distill.item(), emb=emb.item(), target=target.item(),
               score=score.detach().numpy(), hist_vec=hv.numpy(), cand_vec=cv.numpy(), user_vec=uv.numpy(),
               in_hist=hist, in_mask=mask, in_cand=cand, in_label=label)
    for i in range(T):
        rec["in_th%d" % i] = th[i]
        rec["in_tc%d" % i] = tc[i]
    nkeep = 6 if full_grads else 3
    for li, h in enumerate(hidden):
        rec["hidden%d" % li] = h if full_grads else h[:nkeep]
    gnames = []
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.numpy()
        gnames.append(name)
        rec["gnorm." + name] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        if full_grads:
            rec["grad." + name] = g
        else:
            idx, val = grad_samples(seed, name, g)
            rec["gidx." + name] = idx
            rec["gval." + name] = val
    rec["grad_names"] = np.array(gnames)
    rec["meta"] = np.array([seed, B, T, a.user_log_length, a.npratio + 1, L, a.news_dim,
                            cfg_json["num_attention_heads"], a.num_student_layers])
    rec["trainable"] = np.array(sorted(trainable))
    rec["flags"] = np.array([float(a.user_log_mask), a.temperature, a.coef])
    rec["variant"] = np.array([a.pooling, a.model, str(a.num_attention_heads if a.model == "NRMS" else 0)])
    if stats:
        rec["stats"] = np.array([stats])
        rec["hidden_absmax"] = np.array([float(np.abs(h).max()) for h in hidden])
    return rec, P


def golden_relpos(R):
    rel = torch.arange(-511, 512)
    b = R.modeling.relative_position_bucket(rel, num_buckets=32, max_distance=128).numpy()
    out = dict(rel=rel.numpy(), bucket=b)
    w = hashinit.init_tensor(3, "student.news_encoder.bert_model.bert.rel_pos_bias.weight", (12, 32))
    lin = torch.nn.Linear(32, 12, bias=False)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(w))
    for L in (24, 30, 128):
        pos = torch.arange(L)[None]
        rp = pos.unsqueeze(-2) - pos.unsqueeze(-1)
        bk = R.modeling.relative_position_bucket(rp, num_buckets=32, max_distance=128)
        t = lin(torch.nn.functional.one_hot(bk, 32).float()).permute(0, 3, 1, 2)[0]
        out["table%d" % L] = t.detach().numpy()
    out["weight"] = w
    np.savez_compressed(os.path.join(HERE, "relpos.npz"), **out)


def golden_datapath(R):
    d = os.path.join(HERE, "data")
    os.makedirs(d, exist_ok=True)
    rnd = random.Random(5)
    words = "stocks rally as markets open lower after fed signals rate cut storm hits coast team wins final".split()
    with open(os.path.join(d, "news.tsv"), "w") as f:
        for i in range(1, 41):
            n = rnd.randint(0, 3) if i % 13 == 0 else rnd.randint(3, 45 if i % 7 == 0 else 12)
            title = " ".join(rnd.choice(words) for _ in range(n))
            f.write("N%d\tcat%d\tsub%d\t%s\tabs\turl\t[]\t[]\n" % (i, i % 4, i % 9, title))
    lines = []
    for j in range(12):
        nh = [0, 3, 60, 50, 49, 1][j % 6] if j < 6 else rnd.randint(0, 70)
        hist = ["N%d" % rnd.randint(1, 40) for _ in range(nh)]
        if j % 4 == 2 and hist:
            hist[rnd.randrange(len(hist))] = "N999"      # unknown id -> index 0, mask stays 1
        pos = "N%d" % rnd.randint(1, 40)
        negs = ["N%d" % rnd.randint(1, 44) for _ in range(4)]
        lines.append("%d\tU%d\t11/15/2019 8:55:22 AM\t%s\t%s\t%s" % (j, j, " ".join(hist), pos, " ".join(negs)))
    for r in range(3):
        with open(os.path.join(d, "behaviors_np4_%d.tsv" % r), "w") as f:
            f.write("\n".join(lines[r::3]) + "\n")
    a = ref_shim.make_args(num_teachers=2, news_dim=8)
    news, news_index, cat, sub = R.preprocess.read_news_bert(os.path.join(d, "news.tsv"), a, "train")
    nt, nm, _, _ = R.preprocess.get_doc_input_bert(news, news_index, cat, sub, a)
    comb = np.concatenate([nt, nm], -1)
    temb = [hashinit.hash_normal(11, "dp_temb%d" % i, (comb.shape[0], 8)) for i in range(2)]
    dl = R.dataloader.DataLoaderTrain(data_dir=d, filename_pat="behaviors_np4_*.tsv", args=a, world_size=1,
                                      worker_rank=0, cuda_device_idx=0, news_index=news_index,
                                      news_combined=comb, teacher_embs=temb, word_dict=None,
                                      enable_prefetch=False, enable_shuffle=False, enable_gpu=False)
    random.seed(7)
    out = dl._process([l.encode() for l in lines])
    random.seed(7)
    labels = [random.randint(0, a.npratio) for _ in lines]
    rec = dict(news_ids=np.array(list(news_index.keys())), news_index=np.array(list(news_index.values())),
               news_combined=comb, log_ids=out[0].numpy(), log_mask=out[1].numpy(), input_ids=out[2].numpy(),
               targets=out[3].numpy(), labels_seed7=np.array(labels), lines=np.array(lines),
               th0=out[4][0].numpy(), tc1=out[5][1].numpy())
    assert (out[3].numpy() == np.array(labels)).all()
    # sharding rule (streaming.py:40-58); TF reader itself cannot run here
    for w in (1, 2, 3):
        for r in range(w):
            for sh in (False, True):
                fs = R.streaming.get_worker_files(d, r, w, "behaviors_np4_*.tsv", sh, 3)
                rec["files_w%d_r%d_s%d" % (w, r, int(sh))] = np.array([os.path.basename(x) for x in fs])
    np.savez_compressed(os.path.join(HERE, "datapath.npz"), **rec)


def golden_amsgrad():
    rng = np.random.RandomState(0)
    p0 = [rng.randn(5, 7).astype(np.float32), rng.randn(11).astype(np.float32)]
    ps = [torch.nn.Parameter(torch.from_numpy(x.copy())) for x in p0]
    opt = torch.optim.Adam(ps, lr=1e-2, amsgrad=True)
    rec = dict(p0_0=p0[0], p0_1=p0[1])
    for s in range(3):
        gs = [(rng.randn(*x.shape) * (3.0 if s == 0 else 0.3)).astype(np.float32) for x in p0]
        for p, g in zip(ps, gs):
            p.grad = torch.from_numpy(g.copy())
        opt.step()
        for i in range(2):
            rec["g%d_%d" % (s, i)] = gs[i]
            rec["p%d_%d" % (s + 1, i)] = ps[i].detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "amsgrad.npz"), **rec)


def main():
    golden_interface()
    R = ref_shim.load_reference()
    golden_stage1(R)
    golden_convert(R)
    golden_relpos(R)
    golden_datapath(R)
    golden_amsgrad()
    # tiny model: every intermediate + full gradients (pins the oracle's forward AND backward op by op)
    k = 0
    for T, ulm, tau in ((1, False, 1.0), (4, False, 2.0), (4, True, 1.0), (2, True, 2.0)):
        for trainable in ((0, 1), (1,)):
            rec, _ = run_model(R, TINY_CFG, dict(TINY_ARGS, num_student_layers=2, user_log_mask=ulm,
                                                 temperature=tau, coef=0.2), trainable, seed=100 + k, B=4, T=T,
                               full_grads=True)
            np.savez_compressed(os.path.join(HERE, "tiny_model_%d.npz" % k), **rec)
            k += 1
    # full size (H=768, demo.sh shapes U=50 C=5 L=30 D=256): outputs + gradient norms / samples only
    full = [(2, (0, 1), 2, False, 1.0, 21), (4, (2, 3), 4, False, 1.0, 22), (4, (2, 3), 1, True, 2.0, 23)]
    for k, (nl, trainable, T, ulm, tau, seed) in enumerate(full):
        cfg = dict(ref_shim.BASE_CFG, num_hidden_layers=nl)
        rec, _ = run_model(R, cfg, dict(num_student_layers=nl, user_log_mask=ulm, temperature=tau, coef=0.2),
                           trainable, seed=seed, B=2, T=T, full_grads=False)
        np.savez_compressed(os.path.join(HERE, "full_model_%d.npz" % k), **rec)
        print("full", k, rec["total"], rec["distill"], rec["emb"], rec["target"])
    golden_variants(R)
    golden_plmnr()
    golden_stage0(R)
    golden_configs(R)
    golden_pretrained_like(R)
    golden_trajectory(R)


def golden_configs(R):
    """The BASELINE.json configurations that round 1 covered only piecewise, as stated:
    configs[1] PLM-NR 12-layer fine-tune (train 10-11); configs[4] 2-layer student + 4 teachers, stage 2 (titles 30) and
    stage 1 (titles 30 / bodies 128)."""
    golden_plmnr("plmnr_full_1.npz", seed=42, nl=12, trainable=(10, 11))
    cfg = dict(ref_shim.BASE_CFG, num_hidden_layers=2)
    rec, _ = run_model(R, cfg, dict(num_student_layers=2, user_log_mask=False, temperature=1.0, coef=0.2), (0, 1), seed=24,
                       B=2, T=4, full_grads=False)
    np.savez_compressed(os.path.join(HERE, "full_model_5.npz"), **rec)
    print("full 5 (configs[4] stage 2)", rec["total"], rec["distill"], rec["emb"], rec["target"])
    golden_stage1(R, only=("cfg4",))


def golden_trajectory(R, steps=50, n_batches=5, seed=61):
    """A TRAINING TRAJECTORY of the reference (Tiny-NewsRec/run.py:173-200: forward, zero_grad, backward, Adam(amsgrad).step,
    lr 1e-4 as in demo.sh:11): 2-layer student (train 0-1), 2 teachers, B = 2, U = 50, C = 5, L = 30, `steps` steps over a cycle
    of `n_batches` fixed batches.  Committed: the batches, the four losses and the scores of every step (each taken BEFORE that
    step's update, as the loop logs them) and 64 samples of every trainable parameter at the end.  The parity evidence of the other
    goldens stops at two optimiser steps; this one shows the 16-bit engine with its dynamic loss scale following the reference's
    fp32 trajectory."""
    nl, T, B, trainable = 2, 2, 2, (0, 1)
    cfg_json = dict(ref_shim.BASE_CFG, num_hidden_layers=nl)
    a = ref_shim.make_args(config_name=ref_shim.write_config(cfg_json), num_teachers=T, batch_size=B, num_student_layers=nl,
                           user_log_mask=False, temperature=1.0, coef=0.2)
    model = R.model_bert.Model(a)
    fill(model, seed)
    for p in model.teachers.parameters():
        p.requires_grad = False
    for p in model.student.news_encoder.bert_model.parameters():
        p.requires_grad = False
    for i, layer in enumerate(model.student.news_encoder.bert_model.bert.encoder.layer):
        if i in trainable:
            for p in layer.parameters():
                p.requires_grad = True
    L = a.num_words_title
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    batches = [make_inputs(seed + 1 + i, B, a.user_log_length, a.npratio + 1, L, cfg_json["vocab_size"], T, a.news_dim) for i in range(n_batches)]
    lr = 1e-4
    opt = torch.optim.Adam(model.parameters(), lr=lr, amsgrad=True)              # run.py:134
    rec = dict(meta=np.array([seed, B, T, a.user_log_length, a.npratio + 1, L, a.news_dim, cfg_json["num_attention_heads"], nl]),
               trainable=np.array(sorted(trainable)), flags=np.array([0.0, 1.0, 0.2]), lr=np.array([lr]), steps=np.array([steps]),
               n_batches=np.array([n_batches]))
    for i, (hist, mask, cand, label, th, tc) in enumerate(batches):
        rec.update({"in_hist_b%d" % i: hist, "in_mask_b%d" % i: mask, "in_cand_b%d" % i: cand, "in_label_b%d" % i: label})
        for j in range(T):
            rec["in_th%d_b%d" % (j, i)] = th[j]
            rec["in_tc%d_b%d" % (j, i)] = tc[j]
    losses, scores = np.zeros((steps, 4)), np.zeros((steps, B, a.npratio + 1), np.float32)
    for step in range(steps):
        hist, mask, cand, label, th, tc = batches[step % n_batches]
        total, distill, emb, target, score = model(tt(hist), tt(mask), tt(cand), tt(label), [tt(x) for x in th], [tt(x) for x in tc])
        losses[step] = total.item(), distill.item(), emb.item(), target.item()
        scores[step] = score.detach().numpy()
        opt.zero_grad()
        total.backward()
        opt.step()
        if step % 10 == 0:
            print("trajectory step", step, losses[step])
    rec["losses"], rec["scores"] = losses, scores          # columns: total, distill, emb, target
    names = []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        names.append(n)
        idx, val = grad_samples(seed, "traj." + n, p.detach().numpy())
        rec["widx." + n], rec["wval." + n] = idx, val
    rec["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "trajectory_0.npz"), **rec)
    print("trajectory_0", losses[0], losses[-1])


# ---------------------------------------------------------------------------------------------------------------------------
# quality golden (round 6): a corpus a recommender can LEARN, the reference trained on it by its own loop and evaluated by its
# own test(): what "AUC / nDCG within 0.1 pt of the reference" (BASELINE.json north_star) is checked against on the GPU box.

from quality_corpus import dequantize_delta, quality_corpus, quantize_delta  # noqa: E402


def _quality_model(R, seed, nl, T, B, trainable):
    cfg_json = dict(ref_shim.BASE_CFG, num_hidden_layers=nl)
    a = ref_shim.make_args(config_name=ref_shim.write_config(cfg_json), num_teachers=T, batch_size=B, num_student_layers=nl,
                           user_log_mask=True, temperature=1.0, coef=0.2)
    model = R.model_bert.Model(a)
    P0 = fill(model, seed)
    for p in model.teachers.parameters():                                         # run.py:101-112
        p.requires_grad = False
    for p in model.student.news_encoder.bert_model.parameters():
        p.requires_grad = False
    for i, layer in enumerate(model.student.news_encoder.bert_model.bert.encoder.layer):
        if i in trainable:
            for p in layer.parameters():
                p.requires_grad = True
    return model, a, P0, cfg_json


def _quality_eval(R, model, a, c, B):
    """run.py:219-379 on `model`: -> dict(news_scoring, user_vecs, scores, score_offsets, per_impression, metrics)."""
    import importlib
    comb = c["news_combined"]
    modes = [(m, m.training) for m in model.modules()]       # per module: the UniLM inside is in eval mode while the model trains
    model.eval()                                             # (ref_shim: from_pretrained -> .eval(); run.py never calls .train())
    torch.set_grad_enabled(False)
    try:
        scoring = np.concatenate([model.student.news_encoder(torch.from_numpy(comb[i:i + 4 * B])).numpy()
                                  for i in range(0, comb.shape[0], 4 * B)], 0)                       # run.py:276-287
        dt = R.dataloader.DataLoaderTest(data_dir=".", filename_pat="x", args=a, world_size=1, worker_rank=0, cuda_device_idx=0,
                                         news_index=c["news_index"], news_scoring=scoring, word_dict=None, enable_prefetch=False,
                                         enable_shuffle=False, enable_gpu=False)
        sys.path.insert(0, R.path)
        try:
            sys.modules.pop("metrics", None)
            RM = importlib.import_module("metrics")                                                  # the reference's own metrics.py
        finally:
            sys.path.remove(R.path)
            sys.modules.pop("metrics", None)
        tl = [l.encode() for l in c["test_lines"]]
        per, scores, users = [], [], []
        for i in range(0, len(tl), B):
            log_vecs, log_mask, news_vecs, labs = dt._process(tl[i:i + B])
            uv = model.student.user_encoder(log_vecs, log_mask).numpy()                                # run.py:343
            for u, nv, lab in zip(uv, news_vecs, labs):
                sc = np.dot(nv, u)
                scores.append(sc.astype(np.float32))
                users.append(u)
                if lab.mean() == 0 or lab.mean() == 1:                                                 # run.py:346
                    per.append([np.nan] * 4)
                    continue
                per.append([RM.roc_auc_score(lab, sc), RM.mrr_score(lab, sc), RM.ndcg_score(lab, sc, k=5), RM.ndcg_score(lab, sc, k=10)])
    finally:
        torch.set_grad_enabled(True)
        for m, flag in modes:                                # NOT model.train(was_training): that would switch the UniLM's dropout on
            m.training = flag
    per = np.array(per)
    return dict(news_scoring=scoring.astype(np.float32), user_vecs=np.array(users, np.float32),
                score_offsets=np.concatenate([[0], np.cumsum([len(x) for x in scores])]), scores=np.concatenate(scores),
                per_impression=per, metrics=np.nanmean(per, 0))          # AUC, MRR, nDCG@5, nDCG@10 over the impressions run.py scores


def golden_quality(R, steps=400, B=8, lr=1e-4, seed=71, nl=2, T=2, trainable=(0, 1), out_name="quality_0.npz", save=True, requantize=0,
                   save_weights=1, eval_every=0):
    """The reference TRAINED by its own loop (Tiny-NewsRec/run.py:173-200: forward, zero_grad, backward, Adam(amsgrad).step;
    freeze policy :101-112) on the learnable corpus of quality_corpus.py, batches decoded by the reference's own
    DataLoaderTrain._process (dataloader.py:118-172; label draws from random.seed(seed)), then EVALUATED the way run.py:219-379
    does it: every news through student.news_encoder (:276-287), the reference's DataLoaderTest._process per batch
    (dataloader.py:282-310), user vectors through student.user_encoder (:343), np.dot scores and the reference's own metrics.py on
    every impression with both labels present (:346-361).  Committed: the corpus (token table, teacher tables, behaviors
    lines), the label draws, every step's four losses, the per-impression scores and metrics at the end, and the trained
    parameters as int8 DELTAS from the hash init (quantize_delta) - the evaluation runs on exactly W0 + dequantised delta,
    reloaded into the reference model before it is evaluated, so the engine is held to the reference's numbers on bit-identical
    weights; the metrics of the unquantised trained model are kept beside them (`metrics_unquantised`).
    Reproducibility (checked in round 6 by running this function again in the build container): the label draws and all 400 x 4
    losses come out bit-identical; the trained weights do not - torch's CPU backward sums in an order that depends on the thread
    count the run happens to get (8 here, fewer beside other jobs) - so a second run's int8 deltas differ in their last step,
    its metrics by <= 1e-4 (0.009 / 0.0007 / 0 / 0.001 pt) and its per-impression scores by <= 1.1e-3.  The committed file is ONE
    such run; the tests hold the engine to the numbers of exactly the weights it contains.
    requantize=1: take the trained parameters from an existing fixture's fp16 deltas instead of training again.
    save_weights=0, eval_every=k (round 6, `quality_long.npz`: python make_golden.py quality steps=1600 out_name=quality_long.npz
    save_weights=0 eval_every=200): a LONGER run of the same loop on the same corpus without the 15 MB of weight deltas - the
    reference's metrics every k steps (`metrics_at`, its own test() flow each time) and at the end, for the engine-trained
    comparison near convergence (tests/test_quality_gpu.py (iii)); the first 400 steps are quality_0's."""
    import time
    c = quality_corpus(seed, T=T, n_train=steps * B)
    model, a, P0, cfg_json = _quality_model(R, seed, nl, T, B, trainable)
    comb = c["news_combined"]
    out_path = os.path.join(HERE, out_name)
    metrics_at = {}
    if requantize:
        old = np.load(out_path)
        losses, accs, labels = old["losses"], old["accs"], old["labels"]
        assert int(old["steps"][0]) == steps and float(old["lr"][0]) == lr
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.requires_grad:
                    p.copy_(torch.from_numpy(P0[n] + old["delta." + n].astype(np.float32)))
    else:
        dl = R.dataloader.DataLoaderTrain(data_dir=".", filename_pat="x", args=a, world_size=1, worker_rank=0, cuda_device_idx=0,
                                          news_index=c["news_index"], news_combined=comb, teacher_embs=c["tables"], word_dict=None,
                                          enable_prefetch=False, enable_shuffle=False, enable_gpu=False)
        opt = torch.optim.Adam(model.parameters(), lr=lr, amsgrad=True)              # run.py:134
        random.seed(seed)
        lines = [l.encode() for l in c["train_lines"]]
        assert len(lines) >= steps * B
        losses, accs, labels = np.zeros((steps, 4)), np.zeros(steps), np.zeros((steps, B), np.int64)
        t0 = time.time()
        for step in range(steps):
            log_ids, log_mask, input_ids, targets, th, tc = dl._process(lines[step * B:(step + 1) * B])
            total, distill, emb, target, y = model(log_ids, log_mask, input_ids, targets, th, tc)
            losses[step] = total.item(), distill.item(), emb.item(), target.item()
            accs[step] = R.utils.acc(targets, y).item()
            labels[step] = targets.numpy()
            opt.zero_grad()
            total.backward()
            opt.step()
            if eval_every and (step + 1) % eval_every == 0 and step + 1 < steps:
                metrics_at[step + 1] = _quality_eval(R, model, a, c, B)["metrics"]
                print("quality eval at step %d: %s" % (step + 1, np.round(metrics_at[step + 1], 4)), flush=True)
            if step % 10 == 0:
                print("quality step %d  %.1f s  losses %s  acc(last 10) %.3f" % (step, time.time() - t0, np.round(losses[step], 4), accs[max(0, step - 9):step + 1].mean()), flush=True)
    ev_full = _quality_eval(R, model, a, c, B)
    rec = dict(meta=np.array([seed, B, T, a.user_log_length, a.npratio + 1, a.num_words_title, a.news_dim, cfg_json["num_attention_heads"], nl]),
               trainable=np.array(sorted(trainable)), flags=np.array([1.0, 1.0, 0.2]), lr=np.array([lr]), steps=np.array([steps]),
               news_combined=comb.astype(np.int16), train_lines=np.array(c["train_lines"][:steps * B]), test_lines=np.array(c["test_lines"]),
               labels=labels, losses=losses, accs=accs, metrics_unquantised=ev_full["metrics"])
    for i, t in enumerate(c["tables"]):
        rec["table%d" % i] = t
    if metrics_at:
        rec["metrics_at_steps"] = np.array(sorted(metrics_at))
        rec["metrics_at"] = np.stack([metrics_at[k] for k in sorted(metrics_at)])
    if not save_weights:
        rec.update(ev_full)
        print("quality (no weights kept): %d steps  loss %.4f -> %.4f | eval: %s" % (steps, losses[:20, 0].mean(), losses[-20:, 0].mean(),
                                                                                   np.round(ev_full["metrics"], 4)))
        if save:
            np.savez_compressed(out_path, **rec)
        return rec
    # trained parameters -> int8 deltas -> back into the model: what is evaluated below is what is committed
    names, num, den = [], 0.0, 0.0
    with torch.no_grad():
        for n, p in model.named_parameters():
            if not p.requires_grad:
                continue
            d = p.detach().numpy() - P0[n]
            q, scale = quantize_delta(d)
            dq = dequantize_delta(q, scale)
            num, den = num + float(((dq - d).astype(np.float64) ** 2).sum()), den + float((d.astype(np.float64) ** 2).sum())
            names.append(n)
            rec["dq." + n], rec["ds." + n] = q, scale
            p.copy_(torch.from_numpy(P0[n] + dq))
    rec["param_names"] = np.array(names)
    rec["delta_quantisation_rel_l2"] = np.array([np.sqrt(num / den)])
    rec.update(_quality_eval(R, model, a, c, B))
    per = rec["per_impression"]
    print("quality: %d steps  loss %.4f -> %.4f  train acc first / last 20: %.3f / %.3f  |  eval on %d impressions: AUC %.4f MRR %.4f nDCG@5 "
          "%.4f nDCG@10 %.4f  (unquantised weights: %s; int8 deltas are off by %.2f %% of the update's L2)" % (
              steps, losses[:20, 0].mean(), losses[-20:, 0].mean(), accs[:20].mean(), accs[-20:].mean(), int(np.isfinite(per[:, 0]).sum()),
              *rec["metrics"], np.round(ev_full["metrics"], 4), 100 * rec["delta_quantisation_rel_l2"][0]))
    if save:
        np.savez_compressed(out_path, **rec)
    return rec


def golden_pretrained_like(R):
    """The headline model (4 layers, train 2-3, 4 teachers) on weights with a pretrained checkpoint's statistics
    (hashinit.pretrained_like: LayerNorm gamma outliers x 12 ... 30, large embedding rows, |h| ~ 100): what the fp16 build's 16-bit
    activations and loss-scaled backward have to hold up on."""
    cfg = dict(ref_shim.BASE_CFG, num_hidden_layers=4)
    rec, _ = run_model(R, cfg, dict(num_student_layers=4, user_log_mask=False, temperature=1.0, coef=0.2), (2, 3), seed=25,
                       B=2, T=4, full_grads=False, stats="pretrained_like")
    np.savez_compressed(os.path.join(HERE, "full_model_6.npz"), **rec)
    print("full 6 (pretrained-like statistics)", rec["total"], rec["distill"], rec["emb"], rec["target"], "hidden |max|", rec["hidden_absmax"])


def golden_variants(R):
    """pooling in {cls, mean} (model_bert.py:130-135) and the NRMS user encoder (:37-100, :145-148)."""
    tiny = [("cls", "NAML", False), ("mean", "NAML", True), ("att", "NRMS", False), ("att", "NRMS", True),
            ("mean", "NRMS", False)]
    for k, (pool, model, ulm) in enumerate(tiny):
        over = dict(TINY_ARGS, num_student_layers=2, user_log_mask=ulm, temperature=1.0, coef=0.2, pooling=pool,
                    model=model, num_attention_heads=2)
        rec, _ = run_model(R, TINY_CFG, over, (0, 1), seed=200 + k, B=4, T=2, full_grads=True)
        np.savez_compressed(os.path.join(HERE, "tiny_model_%d.npz" % (8 + k)), **rec)
        print("tiny variant", 8 + k, pool, model, ulm, rec["total"])
    full = [("mean", "NRMS", False, 31), ("cls", "NRMS", True, 32)]
    for k, (pool, model, ulm, seed) in enumerate(full):
        cfg = dict(ref_shim.BASE_CFG, num_hidden_layers=2)
        rec, _ = run_model(R, cfg, dict(num_student_layers=2, user_log_mask=ulm, temperature=1.0, coef=0.2, pooling=pool,
                                        model=model, num_attention_heads=16), (0, 1), seed=seed, B=2, T=2, full_grads=False)
        np.savez_compressed(os.path.join(HERE, "full_model_%d.npz" % (3 + k)), **rec)
        print("full variant", 3 + k, pool, model, ulm, rec["total"], rec["distill"], rec["emb"], rec["target"])



def _notebook_distill_classes(R, cfg_json):
    """Execute the class / loss cells of the reference's Post-train_KD.ipynb (cells 11-14) in a namespace that
    resolves their globals; only the published list*tensor bug of cell 14:41 is patched (torch.stack, as in
    model_bert.py:300)."""
    import json
    import torch.nn.functional as Fn
    nb = json.load(open(os.path.join(ref_shim.REF_ROOT, "Post-train_KD.ipynb")))
    tmp = os.path.dirname(ref_shim.write_config(cfg_json))
    os.replace(os.path.join(tmp, "config.json"), os.path.join(tmp, "unilm2-base-uncased-config.json"))
    ns = dict(torch=torch, nn=torch.nn, F=Fn, np=np, os=os, MODEL_CLASSES=R.utils.MODEL_CLASSES, path_turing=tmp)
    for i in (11, 12, 13, 14):
        src = "".join(nb["cells"][i]["source"])
        src = src.replace("        emb_loss = (teacher_MSEs * teacher_weights)",
                          "        teacher_MSEs = torch.stack(teacher_MSEs, dim=-1)\n        emb_loss = (teacher_MSEs * teacher_weights)")
        exec(compile(src, "Post-train_KD.ipynb cell %d" % i, "exec"), ns)
    return ns


def _patch_dropouts(bert, p_hidden, p_attn, seed):
    """Replace the forward of every nn.Dropout the encoder runs in train mode (tnlrv3/modeling.py:177 embeddings, :224 attention
    probabilities, transformers BertSelfOutput / BertOutput at :287 / :306) by a multiplication with the counter-based mask of
    oracle/dropout_oracle.py: the reference's OWN modules then decide where a mask acts and how it is scaled, and the oracle /
    the HIP path are checked against that with the same bits.  DistillModel.forward encodes the bodies first, then the titles
    (Post-train_KD.ipynb cell 12): the n-th call of a module is pass 1 (body) for even n, pass 0 (title) for odd n, forward call
    number = 2 * (n // 2) + pass."""
    from oracle import dropout_oracle as DO
    sites = {"embeddings.dropout": (DO.KIND_EMB, 0)}
    for l in range(len(bert.encoder.layer)):
        sites["encoder.layer.%d.attention.self.dropout" % l] = (DO.KIND_PROB, l)
        sites["encoder.layer.%d.attention.output.dropout" % l] = (DO.KIND_ATTN_OUT, l)
        sites["encoder.layer.%d.output.dropout" % l] = (DO.KIND_FFN_OUT, l)
    mods = dict(bert.named_modules())
    seen = []
    for name, (kind, layer) in sites.items():
        m = mods[name]
        assert isinstance(m, torch.nn.Dropout), name
        state = {"n": 0}

        def fwd(x, kind=kind, layer=layer, state=state, m=m):
            assert m.training
            n = state["n"]
            state["n"] += 1
            call = 2 * (n // 2) + (1 if n % 2 == 0 else 0)
            site = DO.site_id(kind, layer)
            if kind == DO.KIND_PROB:
                N, A, L, _ = x.shape
                mask = DO.probs_mask(p_attn, seed, site, call, N, A, L)
            else:
                N, L, H = x.shape
                mask = DO.rows_mask(p_hidden, seed, site, call, N * L, H).reshape(N, L, H)
            return x * torch.from_numpy(mask)
        m.forward = fwd
        seen.append(name)
    return seen


def golden_stage1_dropout(R):
    """Stage 1 under .train() (Post-train_KD.ipynb cell 19:6) with the masks of oracle/dropout_oracle.py: a tiny case with every
    gradient and the BASELINE configs[4] shapes."""
    golden_stage1(R, only=("tiny", "cfg4"), dropout=(0.1, 0.1, 777))


def golden_stage1(R, only=None, dropout=None):
    import types
    cases = [("tiny", dict(ref_shim.BASE_CFG, hidden_size=64, num_attention_heads=4, intermediate_size=256, vocab_size=128,
                           max_position_embeddings=64, num_hidden_layers=2), dict(news_dim=32, news_query_vector_dim=16),
              (0, 1), 3, 4, 3, 10, 40, 128, True, 300),
             ("full", dict(ref_shim.BASE_CFG, num_hidden_layers=2), dict(news_dim=256, news_query_vector_dim=200),
              (0, 1), 2, 2, 4, 24, 128, 30522, False, 301),
             # BASELINE configs[4] exactly: 2-layer student + FOUR teachers, titles of 30 / bodies of 128 tokens, 1+4 titles
             ("cfg4", dict(ref_shim.BASE_CFG, num_hidden_layers=2), dict(news_dim=256, news_query_vector_dim=200),
              (0, 1), 4, 2, 5, 30, 128, 30522, False, 303)]
    for name, cfg_json, dims, trainable, T, B, C, Lt, Lb, vocab, full, seed in cases:
        if only is not None and name not in only:
            continue
        ns = _notebook_distill_classes(R, cfg_json)
        args = types.SimpleNamespace(num_hidden_layers=cfg_json["num_hidden_layers"], num_teachers=T, **dims)
        model = ns["DistillModel"](args)
        fill(model, seed)
        if dropout is not None:
            model.train()
            patched = _patch_dropouts(model.student.news_encoder.bert_model.bert, *dropout)
            others = [k for k, m in model.named_modules() if isinstance(m, torch.nn.Dropout) and
                      k.replace("student.news_encoder.bert_model.bert.", "") not in patched]
            # anything else must not sit on the path (the dead pooler / classifier head's dropout is never reached)
            assert all(k.endswith("bert_model.dropout") for k in others), others
        for p in model.student.news_encoder.bert_model.parameters():
            p.requires_grad = False
        for i, layer in enumerate(model.student.news_encoder.bert_model.bert.encoder.layer):
            if i in trainable:
                for p in layer.parameters():
                    p.requires_grad = True
        D = dims["news_dim"]

        def toks(tag, n, L):
            ln = hashinit.hash_randint(seed, tag + "len", (n,), 3, L + 1)
            ids = hashinit.hash_randint(seed, tag + "ids", (n, L), 1, vocab)
            m = (np.arange(L)[None, :] < ln[:, None]).astype(np.int64)
            return np.concatenate([ids * m, m], 1)
        title = toks("t", B * C, Lt).reshape(B, C, 2 * Lt)
        body = toks("b", B, Lb)
        label = hashinit.hash_randint(seed, "lab", (B,), 0, C)
        tt = [hashinit.hash_normal(seed, "tt%d" % i, (B, C, D), std=0.3) for i in range(T)]
        tb = [hashinit.hash_normal(seed, "tb%d" % i, (B, D), std=0.3) for i in range(T)]
        t_ = lambda x: torch.from_numpy(np.ascontiguousarray(x))
        loss, target, distill, emb, score = model(t_(title), t_(body), t_(label), [t_(x) for x in tt], [t_(x) for x in tb])
        loss.backward()
        rec = dict(total=loss.item(), target=target.item(), distill=distill.item(), emb=emb.item(), score=score.detach().numpy(),
                   in_title=title, in_body=body, in_label=label,
                   meta=np.array([seed, B, T, C, Lt, Lb, D, cfg_json["num_attention_heads"], cfg_json["num_hidden_layers"]]),
                   trainable=np.array(sorted(trainable)))
        if dropout is not None:
            rec["dropout"] = np.array(dropout, dtype=np.float64)      # p_hidden, p_attn, seed
        for i in range(T):
            rec["in_tt%d" % i], rec["in_tb%d" % i] = tt[i], tb[i]
        gn = []
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.numpy()
            gn.append(k)
            rec["gnorm." + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            if full:
                rec["grad." + k] = g
            else:
                idx, val = grad_samples(seed, k, g)
                rec["gidx." + k], rec["gval." + k] = idx, val
        rec["grad_names"] = np.array(gn)
        np.savez_compressed(os.path.join(HERE, "stage1_%s%s.npz" % (name, "_drop" if dropout is not None else "")), **rec)
        print("stage1", name, "dropout" if dropout is not None else "", rec["total"], rec["target"], rec["distill"], rec["emb"])


from helpers import unilm_checkpoint as _unilm_checkpoint  # noqa: E402


def golden_convert(R):
    """tnlrv3/convert_state_dict.py:load_model and the position-embedding resize of tnlrv3/modeling.py:90-118, run
    from the reference: converted key set + tensors of a small unilm2-layout checkpoint (inputs are regenerated from
    the hash seed by the test), and the resized tables for grow / grow+reuse / shrink."""
    C = R.convert_state_dict
    dims = dict(seed=77, H=16, n_layers=3, A=2, I=32, vocab=40, max_pos=6)
    out = {"dims": np.array([dims[k] for k in ("seed", "H", "n_layers", "A", "I", "vocab", "max_pos")])}
    conv = C.load_model(_unilm_checkpoint(**dims))
    out["keys"] = np.array(sorted(conv))
    for k, v in conv.items():
        out["v." + k] = v.numpy()
    import transformers.models.bert.modeling_bert as mb
    keep = mb.BertPreTrainedModel.__dict__.get("from_pretrained")
    mb.BertPreTrainedModel.from_pretrained = classmethod(lambda cls, *a, **k: k)      # capture what would be loaded
    try:
        for tag, new, reuse in (("grow", 16, None), ("grow_reuse", 16, True), ("shrink", 4, None), ("same", 6, None)):
            cfg = types.SimpleNamespace(max_position_embeddings=new, initializer_range=0.02)
            sd = C.load_model(_unilm_checkpoint(**dims))
            torch.manual_seed(3)
            k = R.modeling.TuringNLRv3PreTrainedModel.from_pretrained.__func__(
                R.modeling.TuringNLRv3ForSequenceClassification, "unused", reuse_position_embedding=reuse, config=cfg,
                state_dict=sd)
            out["pos." + tag] = k["state_dict"]["bert.embeddings.position_embeddings.weight"].numpy()
    finally:
        if keep is None:
            del mb.BertPreTrainedModel.from_pretrained
        else:
            mb.BertPreTrainedModel.from_pretrained = keep
    np.savez_compressed(os.path.join(HERE, "convert.npz"), **out)
    print("convert.npz:", len(conv), "keys")


def golden_plmnr(out_name="plmnr_full_0.npz", seed=41, nl=2, trainable=(0, 1)):
    """BASELINE.json configs[0]/[1]: PLM-NR's ModelBert (PLM-NR/model_bert.py:178-207: same encoders, plain CE) with the
    freeze policy and the two-learning-rate AMSGrad of PLM-NR/run.py:84-106, two optimiser steps.
    plmnr_full_0 = 2 layers (configs[0]'s model); plmnr_full_1 = configs[1] exactly: 12 layers, layers 10-11 trainable
    (PLM-NR/demo.sh:21-22), U=50, C=5, L=30."""
    R = ref_shim.load_reference("PLM-NR")
    B = 2
    cfg_json = dict(ref_shim.BASE_CFG, num_hidden_layers=nl)
    a = ref_shim.make_args(config_name=ref_shim.write_config(cfg_json), num_hidden_layers=nl, batch_size=B)
    model = R.model_bert.ModelBert(a)
    sd = model.state_dict()
    with torch.no_grad():     # same hash weights as the Tiny-NewsRec cases: keys differ by the "student." prefix only
        for k, v in sd.items():
            v.copy_(torch.from_numpy(hashinit.init_tensor(seed, "student." + k, tuple(v.shape))))
    for p in model.news_encoder.bert_model.parameters():
        p.requires_grad = False
    for i, layer in enumerate(model.news_encoder.bert_model.bert.encoder.layer):
        if i in trainable:
            for p in layer.parameters():
                p.requires_grad = True
    inp = make_inputs(seed, B, a.user_log_length, a.npratio + 1, a.num_words_title, cfg_json["vocab_size"], 0, a.news_dim)
    hist, mask, cand, label = inp[:4]
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    bert = [p for n, p in model.named_parameters() if ".bert_model." in n]
    rest = [p for n, p in model.named_parameters() if ".bert_model." not in n]
    lr_bert, lr = 1e-5, 1e-4
    opt = torch.optim.Adam([{"params": bert, "lr": lr_bert}, {"params": rest, "lr": lr}], amsgrad=True)
    rec = dict(in_hist=hist, in_mask=mask, in_cand=cand, in_label=label, lrs=np.array([lr_bert, lr]),
               meta=np.array([seed, B, 0, a.user_log_length, a.npratio + 1, a.num_words_title, a.news_dim,
                              cfg_json["num_attention_heads"], nl]), trainable=np.array(trainable))
    names = []
    for step in range(2):
        opt.zero_grad()
        loss, score = model(tt(hist), tt(mask), tt(cand), tt(label))
        loss.backward()
        rec["loss%d" % step] = loss.item()
        rec["score%d" % step] = score.detach().numpy()
        if step == 0:
            for n, p in model.named_parameters():
                if p.grad is None:
                    continue
                g = p.grad.numpy()
                names.append(n)
                rec["gnorm." + n] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                idx, val = grad_samples(seed, n, g)
                rec["gidx." + n], rec["gval." + n] = idx, val
        opt.step()
    rec["grad_names"] = np.array(names)
    for n in ("news_encoder.dense.weight", "news_encoder.bert_model.bert.encoder.layer.%d.output.dense.weight" % max(trainable),
              "user_encoder.attn.att_fc1.weight"):
        w = dict(model.named_parameters())[n].detach().numpy()
        idx, val = grad_samples(seed, "w." + n, w)
        rec["widx." + n], rec["wval." + n] = idx, val            # parameter samples after the two steps
    np.savez_compressed(os.path.join(HERE, out_name), **rec)
    print(out_name, rec["loss0"], rec["loss1"])


def golden_plmnr_hf(model_type, seed):
    """PLM-NR's ModelBert with --model_type bert / roberta (PLM-NR/utils.py:17-21: transformers BertModel / RobertaModel as the
    news encoder, PLM-NR/model_bert.py:109-118, hidden state of the last layer): 2 layers, both trainable (PLM-NR/run.py:119-124
    unfreezes bert_model.encoder.layer[i]), one forward / backward.  Weights under the reference's own key names
    ("student." + key for the shared hash generator); RoBERTa inputs use its padding id 1 and its config (type_vocab 1,
    514 positions, layer_norm_eps 1e-5)."""
    import transformers
    R = ref_shim.load_reference("PLM-NR")
    cls = {"bert": transformers.BertModel, "roberta": transformers.RobertaModel}[model_type]
    keep = cls.__dict__.get("from_pretrained")
    cls.from_pretrained = classmethod(lambda c, path, config=None, **kw: c(config).eval())
    try:
        nl, B = 2, 2
        cfg_json = dict(ref_shim.BASE_CFG, num_hidden_layers=nl)
        cfg_json.pop("rel_pos_bins"), cfg_json.pop("max_rel_pos")
        if model_type == "roberta":
            cfg_json.update(pad_token_id=1, type_vocab_size=1, max_position_embeddings=514, layer_norm_eps=1e-5, vocab_size=50265)
        cfg_json["_attn_implementation"] = "eager"
        a = ref_shim.make_args(config_name=ref_shim.write_config(cfg_json), num_hidden_layers=nl, batch_size=B, model_type=model_type)
        model = R.model_bert.ModelBert(a)
        sd = model.state_dict()
        with torch.no_grad():
            for k, v in sd.items():
                v.copy_(torch.from_numpy(hashinit.init_tensor(seed, "student." + k, tuple(v.shape))))
        for p_ in model.news_encoder.bert_model.parameters():
            p_.requires_grad = False
        for layer in model.news_encoder.bert_model.encoder.layer:
            for p_ in layer.parameters():
                p_.requires_grad = True
        inp = make_inputs(seed, B, a.user_log_length, a.npratio + 1, a.num_words_title, 30522, 0, a.news_dim)
        hist, mask, cand, label = [np.array(x) for x in inp[:4]]
        if model_type == "roberta":          # its tokenizer pads with id 1 and never emits 0 / 1 inside a text
            L = a.num_words_title
            for x in (hist, cand):
                ids, m = x[..., :L], x[..., L:]
                x[..., :L] = np.where(m > 0, np.maximum(ids, 2), 1)
        tt = lambda x: torch.from_numpy(np.ascontiguousarray(x))
        loss, score = model(tt(hist), tt(mask), tt(cand), tt(label))
        loss.backward()
        assert torch.isfinite(loss) and torch.isfinite(score).all()
        rec = dict(in_hist=hist, in_mask=mask, in_cand=cand, in_label=label, loss0=loss.item(), score0=score.detach().numpy(),
                   meta=np.array([seed, B, 0, a.user_log_length, a.npratio + 1, a.num_words_title, a.news_dim,
                                  cfg_json["num_attention_heads"], nl]), trainable=np.array([0, 1]),
                   keys=np.array(sorted(k for k in sd if not k.endswith("position_ids"))),
                   dims=np.array([cfg_json["vocab_size"], cfg_json["max_position_embeddings"], cfg_json["type_vocab_size"]]),
                   ln_eps=np.float64(cfg_json.get("layer_norm_eps", 1e-12)))
        names = []
        for n, p_ in model.named_parameters():
            if p_.grad is None:
                continue
            g = p_.grad.numpy()
            names.append(n)
            rec["gnorm." + n] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            idx, val = grad_samples(seed, n, g)
            rec["gidx." + n], rec["gval." + n] = idx, val
        rec["grad_names"] = np.array(names)
        np.savez_compressed(os.path.join(HERE, "plmnr_%s.npz" % model_type), **rec)
        print("plmnr_%s.npz" % model_type, rec["loss0"], len(names), "gradients")
    finally:
        if keep is None:
            del cls.from_pretrained
        else:
            cls.from_pretrained = keep


def golden_stage0(R):
    """Domian-specific_Post-train.ipynb (stage 0: contrastive title/body matching of the teacher): its
    TitleBodySimModel (cells 10-11) executed as published -- CE over 1+K title scores per body, no teachers."""
    import json
    import types
    nb = json.load(open(os.path.join(ref_shim.REF_ROOT, "Domian-specific_Post-train.ipynb")))
    cfg_json = dict(ref_shim.BASE_CFG, num_hidden_layers=12)       # the notebook hard-codes 12 layers (cell 10)
    tmp = os.path.dirname(ref_shim.write_config(cfg_json))
    os.replace(os.path.join(tmp, "config.json"), os.path.join(tmp, "unilm2-base-uncased-config.json"))
    ns = dict(torch=torch, nn=torch.nn, np=np, os=os, MODEL_CLASSES=R.utils.MODEL_CLASSES, path_turing=tmp)
    for i in (10, 11):
        exec(compile("".join(nb["cells"][i]["source"]), "Domian-specific_Post-train.ipynb cell %d" % i, "exec"), ns)
    seed, B, C, Lt, Lb, D, vocab = 302, 2, 4, 24, 96, 256, 30522
    trainable = (9, 10, 11)                                        # cell 14
    model = ns["TitleBodySimModel"](types.SimpleNamespace(news_dim=D, news_query_vector_dim=200))
    sd = model.state_dict()
    with torch.no_grad():     # hash weights under the Tiny-NewsRec key names the build's engine uses
        for k, v in sd.items():
            v.copy_(torch.from_numpy(hashinit.init_tensor(seed, "student." + k, tuple(v.shape))))
    for p in model.news_encoder.bert_model.parameters():
        p.requires_grad = False
    for i, layer in enumerate(model.news_encoder.bert_model.bert.encoder.layer):
        if i in trainable:
            for p in layer.parameters():
                p.requires_grad = True

    def toks(tag, n, L):
        ln = hashinit.hash_randint(seed, tag + "len", (n,), 3, L + 1)
        ids = hashinit.hash_randint(seed, tag + "ids", (n, L), 1, vocab)
        m = (np.arange(L)[None, :] < ln[:, None]).astype(np.int64)
        return np.concatenate([ids * m, m], 1)
    title = toks("t", B * C, Lt).reshape(B, C, 2 * Lt)
    body = toks("b", B, Lb)
    label = hashinit.hash_randint(seed, "lab", (B,), 0, C)
    t_ = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    score, loss = model(t_(title), t_(body), t_(label))
    loss.backward()
    rec = dict(total=loss.item(), score=score.detach().numpy(), in_title=title, in_body=body, in_label=label,
               meta=np.array([seed, B, 0, C, Lt, Lb, D, 12, 12]), trainable=np.array(trainable))
    gn = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.numpy()
        gn.append(k)
        rec["gnorm." + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        rec["gidx." + k], rec["gval." + k] = grad_samples(seed, k, g)
    rec["grad_names"] = np.array(gn)
    np.savez_compressed(os.path.join(HERE, "stage0_full.npz"), **rec)
    print("stage0", rec["total"], len(gn), "gradients")


def golden_interface():
    """Flag surface (parameters.py) and state_dict schema (model_bert.Model) of the reference, as JSON."""
    import json
    import importlib
    R = ref_shim.load_reference()
    sys.path.insert(0, R.path)
    try:
        sys.modules.pop("parameters", None)
        P = importlib.import_module("parameters")
        argv, sys.argv = sys.argv, ["x"]
        try:
            a = P.parse_args()
        finally:
            sys.argv = argv
    finally:
        sys.path.remove(R.path)
        sys.modules.pop("parameters", None)
    flags = {k: (v if not isinstance(v, float) else float(v)) for k, v in vars(a).items()}
    m = R.model_bert.Model(ref_shim.make_args(num_student_layers=4, num_teachers=4))
    schema = {k: list(v.shape) for k, v in m.state_dict().items()}
    n_train = 0
    for p in m.teachers.parameters():
        p.requires_grad = False
    for p in m.student.news_encoder.bert_model.parameters():
        p.requires_grad = False
    for i, layer in enumerate(m.student.news_encoder.bert_model.bert.encoder.layer):
        if i in (2, 3):
            for p in layer.parameters():
                p.requires_grad = True
    trainable = [k for k, p in m.named_parameters() if p.requires_grad]
    with open(os.path.join(HERE, "interface.json"), "w") as f:
        json.dump({"flags": flags, "state_dict": schema, "trainable_4layer_23": trainable}, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "configs":       # only the round-2 additions (the other fixtures are unchanged)
        golden_configs(ref_shim.load_reference())
    elif len(sys.argv) > 1 and sys.argv[1] == "hf":
        golden_plmnr_hf("bert", 51)
        golden_plmnr_hf("roberta", 52)
    elif len(sys.argv) > 1 and sys.argv[1] == "trajectory":
        golden_trajectory(ref_shim.load_reference())
    elif len(sys.argv) > 1 and sys.argv[1] == "quality":
        golden_quality(ref_shim.load_reference(), **{k: type(dict(steps=1, B=1, lr=1.0, requantize=0, save_weights=1, eval_every=0, out_name="")[k])(v)
                                                     for k, v in (x.split("=") for x in sys.argv[2:])})
    elif len(sys.argv) > 1 and sys.argv[1] == "dropout":
        golden_stage1_dropout(ref_shim.load_reference())
    else:
        main()
