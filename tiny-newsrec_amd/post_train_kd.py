"""Title/body matching post-training as a script: the flows of the reference's two notebooks on the HIP engine.

  --stage 1 (default)  Post-train_KD.ipynb (cells 2-9 data, 15-19 training): multi-teacher distillation of the student.
  --stage 0            Domian-specific_Post-train.ipynb: the teacher's own contrastive post-training (TitleBodySimModel,
                       cells 10-16: 12 layers, layers 9-11 trainable, plain CE, checkpoint every --save_steps steps) and,
                       with --mode export, its cells 18-22: title / body embeddings of the whole corpus under each of
                       --ckpt_paths, written as teacher_title_emb_{i}.pkl / teacher_body_emb_{i}.pkl for stage 1.

    python post_train_kd.py --corpus_path docs_filter.tsv --teacher_emb_dir . --num_teachers 4 --num_hidden_layers 4
    python -m torch.distributed.run --nproc-per-node 8 ... post_train_kd.py ...        (data-parallel, RCCL)
    python post_train_kd.py --synthetic True --max_steps 20                            (no files needed)

What the notebook does per sample (DistillDataset.__getitem__, cell 8) -- one positive document, NPRATIO negatives
drawn without replacement from the other documents, titles tokenised to 24 and the positive's body to 512
wordpieces, the teachers' title / body embeddings of the same documents, label 0 -- happens here at index level:
titles and bodies are tokenised ONCE into resident (n, 2L) int32 tables, teacher embeddings stay in HBM as (T, n, D)
tables, and a step ships B x (1+K) document indices (stage1.Stage1Engine.forward_indexed).  Freeze policy (cell 17),
two-rate plain Adam (cell 18), log line and checkpoint name / layout (cell 19) are the notebook's.
Data-parallel runs shard the epoch's permutation by rank and average gradients like run.py."""
import argparse
import logging
import os
import pickle
import random
import time

import numpy as np
import torch

import utils


def parse_args(argv=None):
    b = utils.str2bool
    p = argparse.ArgumentParser()
    p.add_argument("--stage", type=int, default=1, choices=[0, 1])
    p.add_argument("--mode", default="train", choices=["train", "export"])
    p.add_argument("--ckpt_paths", nargs="*", default=[], help="--mode export: checkpoints to embed the corpus with")
    p.add_argument("--save_steps", type=int, default=500, help="stage 0: checkpoint period (args.T of the notebook)")
    p.add_argument("--corpus_path", default="./docs_filter.tsv")                 # cell 2
    p.add_argument("--teacher_emb_dir", default=".", help="teacher_title_emb_{i}.pkl / teacher_body_emb_{i}.pkl (cell 7)")
    p.add_argument("--num_teachers", type=int, default=4)
    p.add_argument("--num_hidden_layers", type=int, default=4)
    p.add_argument("--news_dim", type=int, default=256)
    p.add_argument("--news_query_vector_dim", type=int, default=200)
    p.add_argument("--max_title_len", type=int, default=24)                       # cell 4
    p.add_argument("--max_body_len", type=int, default=512)
    p.add_argument("--npratio", type=int, default=9)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--epochs", type=int, default=1)                               # cell 19
    p.add_argument("--bert_trainable_layer", type=int, nargs="+", default=[2, 3])  # cell 17
    p.add_argument("--pretrain_lr", type=float, default=1e-6)                     # cell 18
    p.add_argument("--lr", type=float, default=1e-5)
    p.add_argument("--tokenizer_name", default="./unilmv2/unilm2-base-uncased-vocab.txt")
    p.add_argument("--config_name", default="./unilmv2/unilm2-base-uncased-config.json")
    p.add_argument("--model_name", default="./unilmv2/unilm2-base-uncased.bin")
    p.add_argument("--save_dir", default=".")
    p.add_argument("--log_steps", type=int, default=10)
    p.add_argument("--max_steps", type=int, default=0, help="stop an epoch early (0 = whole corpus)")
    p.add_argument("--seed", type=int, default=0, help="shuffle / sampling / dropout-mask seed; the mask counter restarts at 0 in every process, so a run "
                   "continued from a checkpoint should pass a different seed (as a fixed torch seed would replay the notebook's masks)")
    p.add_argument("--enable_hvd", type=b, default=True, help="data-parallel when launched by torch.distributed.run")
    p.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"])
    p.add_argument("--hidden_dropout_prob", type=float, default=None,
                   help="train-mode dropout (the notebooks call .train(): cell 19:6 / cell 16:6); default = the value in --config_name "
                        "(tnlrv3/config/*.json:4, 0.1), 0 turns it off")
    p.add_argument("--attention_probs_dropout_prob", type=float, default=None, help="same for the attention probabilities (config json:2)")
    p.add_argument("--allow_random_init", type=b, default=False, help="train from the construction-time initialisation when --model_name does not exist")
    p.add_argument("--synthetic", type=b, default=False, help="hash-initialised weights + synthetic corpus / teacher tables")
    p.add_argument("--synthetic_docs", type=int, default=20000)
    a = p.parse_args(argv)
    if a.stage == 0:                                   # Domian-specific_Post-train.ipynb cells 10, 14
        a.num_teachers = 0
        if "--num_hidden_layers" not in (argv or __import__("sys").argv):
            a.num_hidden_layers = 12
        if "--bert_trainable_layer" not in (argv or __import__("sys").argv):
            a.bert_trainable_layer = [9, 10, 11]
    return a


def read_corpus(path):
    """cells 5-6: tab-separated lines, title in column 3, body in column 4."""
    titles, bodies = [], []
    with open(path, encoding="utf-8") as f:
        for line in f:
            cols = line.strip("\n").split("\t")
            titles.append(cols[3])
            bodies.append(cols[4])
    return titles, bodies


def token_table(texts, encode, width):
    """(n, 2*width) int32 rows [input_ids | attention_mask], padded / truncated as the notebook's tokenizer call."""
    out = np.zeros((len(texts), 2 * width), dtype=np.int32)
    for i, text in enumerate(texts):
        tok = encode(text, max_length=width, padding="max_length", truncation=True)
        out[i, :width] = tok["input_ids"]
        out[i, width:] = tok["attention_mask"]
    return out


def load_teacher_tables(args, n_docs):
    tt, tb = [], []
    for i in range(args.num_teachers):
        for name, dst in (("teacher_title_emb_%d.pkl" % i, tt), ("teacher_body_emb_%d.pkl" % i, tb)):
            with open(os.path.join(args.teacher_emb_dir, name), "rb") as f:
                a = np.asarray(pickle.load(f), dtype=np.float32)
            assert a.shape == (n_docs, args.news_dim), "%s: %s, expected %s" % (name, a.shape, (n_docs, args.news_dim))
            dst.append(a)
    return np.stack(tt, 0), np.stack(tb, 0)


def sample_indices(rng, positives, n_docs, k):
    """DistillDataset.__getitem__ (cell 8) at index level: for every positive, k distinct other documents."""
    out = np.empty((len(positives), 1 + k), dtype=np.int32)
    for r, pos in enumerate(positives):
        neg = rng.sample(range(n_docs - 1), k)
        out[r, 0] = pos
        out[r, 1:] = [j + (j >= pos) for j in neg]                # skip the positive itself
    return out


def _model_dims(args):
    from model_bert import read_model_config
    cfg = read_model_config(args.config_name, args.synthetic)
    return dict(hidden=cfg.get("hidden_size", 768), heads=cfg.get("num_attention_heads", 12), inter=cfg.get("intermediate_size", 3072),
                vocab=cfg.get("vocab_size", 30522), max_pos=max(cfg.get("max_position_embeddings", 512), args.max_body_len),
                type_vocab=cfg.get("type_vocab_size", 2), ln_eps=cfg.get("layer_norm_eps", 1e-12),
                news_dim=args.news_dim, news_query=args.news_query_vector_dim)


def _ckpt_state(eng, stage):
    """Checkpoint keys: DistillModel's (student.news_encoder.*, transform_matrix.*) for stage 1, TitleBodySimModel's
    (news_encoder.*) for stage 0."""
    sd = {k: v.cpu() for k, v in eng.state_dict().items()}
    return sd if stage == 1 else {k[len("student."):]: v for k, v in sd.items()}


def _load_ckpt(eng, path, stage):
    sd = torch.load(path, map_location="cpu")
    sd = sd.get("model_state_dict", sd)
    if stage == 0:
        sd = {"student." + k: v for k, v in sd.items()}
    eng.load_state_dict(sd)


def train(args):
    import dist
    from model_bert import load_pretrained_into, reference_init
    from stage1 import Stage1Engine
    size, rank, local = utils.init_hvd_cuda(args.enable_hvd, True)
    dev = "cuda:%d" % local
    random.seed(args.seed + rank)
    if args.synthetic:
        import hashinit
        import synth
        n_docs = args.synthetic_docs
        title_tab = synth.news_table(11, n_docs - 1, args.max_title_len)          # (n_docs, 2Lt), row 0 = empty document
        body_tab = synth.news_table(12, n_docs - 1, args.max_body_len, mean_len=0.6 * args.max_body_len,
                                    std_len=0.25 * args.max_body_len)
        tt = synth.teacher_tables(13, max(args.num_teachers, 1), n_docs - 1, args.news_dim)
        tb = synth.teacher_tables(14, max(args.num_teachers, 1), n_docs - 1, args.news_dim)
    else:
        from preprocess import make_tokenizer
        encode = make_tokenizer(args)
        titles, bodies = read_corpus(args.corpus_path)
        n_docs = len(titles)
        title_tab = token_table([t.lower() for t in titles], encode, args.max_title_len)
        body_tab = token_table([t.lower() for t in bodies], encode, args.max_body_len)
        tt, tb = load_teacher_tables(args, n_docs) if args.num_teachers else (np.zeros((1, 1, args.news_dim), np.float32),) * 2
    if size > 1:
        import engine as E
        E.Engine.WGRAD_UNITS = 2             # see run.py: weight-gradient units beside an overlapped all-reduce
    eng = Stage1Engine(n_layers=args.num_hidden_layers, trainable_layers=[l for l in args.bert_trainable_layer if l < args.num_hidden_layers],
                       num_teachers=args.num_teachers, npratio=args.npratio, title_len=args.max_title_len,
                       body_len=args.max_body_len, device=dev, batch=args.batch_size, dtype=args.dtype, **_model_dims(args))
    from model_bert import read_model_config
    mcfg = read_model_config(args.config_name, args.synthetic)
    p_h = mcfg.get("hidden_dropout_prob", 0.1) if args.hidden_dropout_prob is None else args.hidden_dropout_prob
    p_a = mcfg.get("attention_probs_dropout_prob", 0.1) if args.attention_probs_dropout_prob is None else args.attention_probs_dropout_prob
    if args.mode == "train":
        eng.set_dropout(p_h, p_a, seed=args.seed * 1000003 + rank)       # every rank draws its own masks, like per-process torch RNGs
        logging.info("[%d] train-mode dropout: hidden %.3g, attention probabilities %.3g", rank, p_h, p_a)
    if args.synthetic:
        eng.load_state_dict({k: torch.from_numpy(hashinit.init_tensor(1234, k, tuple(s))) for k, s in eng.shapes.items()})
    else:
        reference_init(eng.title, args.seed)
        rep = load_pretrained_into(eng.title, args.model_name, args.seed, allow_missing=args.allow_random_init)
        logging.info("pretrained encoder: %s", "not found, random init" if rep is None else "%d missing / %d unexpected keys" % (len(rep[0]), len(rep[1])))
        eng.body.refresh_rel()
    t_eng = eng.title
    dist.broadcast_flat([t_eng.flat[True], t_eng.flat[False]])
    t_eng.refresh_shadows(all_layers=True)
    eng.body.refresh_rel()
    sync = dist.GradSync(t_eng.flat_g, eng.bucket_ranges(), size) if size > 1 else None
    d_title, d_body = torch.from_numpy(title_tab).to(dev), torch.from_numpy(body_tab).to(dev)
    if args.mode == "export":                          # Domian-specific_Post-train.ipynb cells 18-22
        os.makedirs(args.save_dir, exist_ok=True)
        for i, ck in enumerate(args.ckpt_paths):
            _load_ckpt(eng, ck, args.stage)
            for which, tab in (("title", d_title), ("body", d_body)):
                emb = eng.encode_table(tab, which).cpu().numpy()
                if rank == 0:
                    with open(os.path.join(args.save_dir, "teacher_%s_emb_%d.pkl" % (which, i)), "wb") as f:
                        pickle.dump(emb, f)
            logging.info("[%d] %s -> teacher_{title,body}_emb_%d.pkl (%d documents)", rank, ck, i, n_docs)
        return eng
    d_tt, d_tb = torch.from_numpy(np.ascontiguousarray(tt)).to(dev), torch.from_numpy(np.ascontiguousarray(tb)).to(dev)
    B = args.batch_size
    label = torch.zeros(B, dtype=torch.int64, device=dev)                          # the positive comes first (cell 8)
    for ep in range(args.epochs):
        order = list(range(n_docs))
        random.Random(args.seed + ep).shuffle(order)                               # DataLoader(shuffle=True), same on every rank
        order = order[rank::size]
        per_rank = (n_docs // size) // B                        # the same on every rank (len(order) differs by one across ranks)
        steps = per_rank if not args.max_steps else min(args.max_steps, per_rank)
        sums = torch.zeros(5, device=dev)                                          # loss, target, distill, emb, acc
        t0 = time.time()
        for cnt in range(1, steps + 1):
            idx_h = sample_indices(random, order[(cnt - 1) * B:cnt * B], n_docs, args.npratio)
            idx = torch.from_numpy(idx_h).to(dev)
            body_idx = torch.from_numpy(np.ascontiguousarray(idx_h[:, 0])).to(dev) if idx_h.dtype == np.int32 else None
            losses, score = eng.forward_indexed(d_title, d_body, idx, label, d_tt, d_tb, body_idx=body_idx)
            sums[0] += eng.total_loss()
            sums[1] += losses[1]
            sums[2] += losses[0]
            sums[3] += losses[2]
            sums[4] += utils.acc(label, score)
            eng.backward(after_bucket=sync.launch if sync else None)
            eng.step(args.lr, grad_scale=sync.scale if sync else 1.0, lr_bert=args.pretrain_lr, amsgrad=False, sync=sync)
            if args.stage == 0 and rank == 0 and cnt % args.save_steps == 0:          # cell 16: DP_12_layer_{cnt}.pt
                os.makedirs(args.save_dir, exist_ok=True)
                torch.save({"model_state_dict": _ckpt_state(eng, 0)},
                           os.path.join(args.save_dir, "DP_%d_layer_%d.pt" % (args.num_hidden_layers, cnt)))
            if cnt % args.log_steps == 0:
                s = (sums / cnt).tolist()
                sc = eng.title.scaler
                logging.info("[%d] ed: %d, loss: %.5f, t_loss: %.5f, d_loss: %.5f, e_loss: %.5f, acc: %.5f, %.1f pairs/s%s" % (
                    rank, cnt * B, s[0], s[1], s[2], s[3], s[4], size * cnt * B / max(time.time() - t0, 1e-9),
                    ", loss scale %g, %d steps skipped (fp16 overflow)" % (eng.title.gscale, sc.skipped) if sc.enabled and sc.skipped else ""))
        if eng.title.scaler.enabled:
            eng.title.scaler.drain(eng.title)
        if rank == 0:
            os.makedirs(args.save_dir, exist_ok=True)
            name = "first_stage_%d_layer.pt" if args.stage == 1 else "DP_%d_layer.pt"
            path = os.path.join(args.save_dir, name % args.num_hidden_layers)
            torch.save({"model_state_dict": _ckpt_state(eng, args.stage)}, path)
            logging.info("Model saved to %s", path)
    return eng


if __name__ == "__main__":
    utils.setuplogger()
    train(parse_args())
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
