"""Entry point with the reference's interface (Tiny-NewsRec/run.py:463-472): python run.py --mode train ...
Training loop = run.py:20-216 (step order fwd -> acc -> zero_grad -> backward -> step, log line format,
epoch-end checkpoint dict and file name, max_steps_per_epoch break), on the HIP engine with RCCL replacing
horovod.  --mode test / get_teacher_emb are the forward-only paths of SURVEY.md section 8-f N2."""
import logging
import os
import sys
import pickle
import time

import numpy as np
import torch

import utils
from parameters import parse_args


def _load_inputs(args):
    """news table + teacher embeddings (run.py:46-53, 73-74) or their synthetic stand-ins."""
    if args.synthetic:
        import synth
        n = 51282
        comb = synth.news_table(1234, n, args.num_words_title)
        tables = list(synth.teacher_tables(1234, args.num_teachers, n, args.news_dim)) if args.num_teachers else []
        return {"N%d" % i: i for i in range(1, n + 1)}, comb, tables, {}, {}
    from preprocess import get_doc_input_bert, read_news_bert
    news, news_index, cat, sub = read_news_bert(os.path.join(args.train_data_dir, "news.tsv"), args, mode="train")
    title, mask, _, _ = get_doc_input_bert(news, news_index, cat, sub, args)
    tables = []
    for p in args.teacher_emb_paths[:args.num_teachers]:
        with open(p, "rb") as f:
            tables.append(np.asarray(pickle.load(f), dtype=np.float32))
    return news_index, np.concatenate([title, mask], axis=-1), tables, cat, sub


def train(args):
    import dist
    from dataloader import DataLoaderTrain, IndexBatch
    from model_bert import Model, ModelBert, TnrAdam
    from streaming import get_stat, get_worker_files
    size, rank, local = utils.init_hvd_cuda(args.enable_hvd, args.enable_gpu)
    assert args.enable_gpu, "the HIP path needs a GPU (there is no CPU fallback)"
    # the loader's producer thread (TSV decode) and this thread (kernel launches) share the GIL: with the default 5 ms switch
    # interval the launching thread can sit out a whole 3 ms step behind the decoder and the GPU queue runs dry
    sys.setswitchinterval(2e-4)
    news_index, news_combined, teacher_embs, category_dict, subcategory_dict = _load_inputs(args)
    # --num_teachers 0 is PLM-NR/run.py's train(): the same encoders with plain CE (ModelBert), checkpoints with PLM-NR's
    # key names, and -- with --use_pretrain_model -- its two learning rates (PLM-NR/run.py:56-106).  Tiny-NewsRec's own
    # Model cannot be built without teachers, so the flag value is free for this.
    plmnr = args.num_teachers == 0
    if size > 1:
        # beside the CUs an overlapped gradient all-reduce holds, a late workgroup of the weight-gradient kernel costs one work
        # unit: two units per workgroup halve that (engine.Engine.WGRAD_UNITS; set before the engine sizes its slab workspace)
        import engine as E
        E.Engine.WGRAD_UNITS = 2
    model = ModelBert(args) if plmnr else Model(args)
    eng = model.engine
    sd = model.state_dict()
    pretrained = False
    if args.synthetic:
        import hashinit
        pfx = "student." if plmnr else ""
        sd = {k: torch.from_numpy(hashinit.init_tensor(1234, pfx + k, tuple(v.shape))) for k, v in sd.items()}
    elif plmnr:
        if args.use_pretrain_model:                                          # PLM-NR/run.py:56-70 (first-stage student -> this model)
            for k, v in torch.load(args.pretrain_model_path, map_location="cpu")["model_state_dict"].items():
                k2 = k[len("student."):] if k.startswith("student.") else None
                if k2 in sd:
                    sd[k2] = v
                    pretrained = True
    else:
        for i, ck in enumerate(args.teacher_ckpts[:args.num_teachers]):      # run.py:61-70
            for k, v in torch.load(ck, map_location="cpu")["model_state_dict"].items():
                if k.startswith("user_encoder"):
                    sd[".".join(["teachers", str(i)] + k.split(".")[1:])] = v
        if args.use_pretrain_model:                                          # run.py:76-85
            for k, v in torch.load(args.pretrain_model_path, map_location="cpu")["model_state_dict"].items():
                if k.startswith("student"):
                    sd[k] = v
    model.load_state_dict(sd)
    if args.load_ckpt_name is not None:                                      # run.py:125-129
        ck = utils.get_checkpoint(args.model_dir, args.load_ckpt_name)
        model.load_state_dict(torch.load(ck, map_location="cpu")["model_state_dict"])
        logging.info(f"Model loaded from {ck}")
    dist.broadcast_flat([eng.flat[True], eng.flat[False]])                   # hvd.broadcast_parameters, run.py:142
    eng.refresh_shadows(all_layers=True)
    sync = dist.GradSync(eng.flat_g, eng.bucket_ranges(), size)             # hvd.DistributedOptimizer(Average), :145-149
    model._after_bucket = sync.launch if size > 1 else None
    optimizer = TnrAdam(model, args.lr, sync if size > 1 else None,
                        pretrain_lr=args.pretrain_lr if (plmnr and pretrained) else None,      # PLM-NR/run.py:94-106
                        pretrained_heads=plmnr and pretrained)

    if args.synthetic:
        import synth
        steps = min(args.max_steps_per_epoch, 200)
        steps_cap = steps
        hidx, mask, cidx, label = [torch.from_numpy(x).cuda() for x in
                                   synth.impressions(77 + rank, steps * args.batch_size, len(news_index), args.user_log_length, args.npratio + 1)]
        B = args.batch_size
        dev_news = torch.from_numpy(news_combined).cuda()
        dev_tab = torch.from_numpy(np.stack(teacher_embs, 0)).cuda() if teacher_embs else None
        plans = [None] * steps
        if args.dedup_news:
            from dedup import build_plan
            hn, cn = hidx.cpu().numpy(), cidx.cpu().numpy()
            plans = [build_plan(hn[i * B:(i + 1) * B], cn[i * B:(i + 1) * B]) for i in range(steps)]
            plans = [p.to("cuda") if p is not None else None for p in plans]
        batches = lambda: (IndexBatch((hidx[i * B:(i + 1) * B], mask[i * B:(i + 1) * B], cidx[i * B:(i + 1) * B],
                                       label[i * B:(i + 1) * B], plans[i])) for i in range(steps))
    else:
        stat = get_stat(args.train_data_dir, args.filename_pat)
        paths = get_worker_files(args.train_data_dir, rank, size, args.filename_pat, args.enable_shuffle, 0)
        n = sum(stat[f] for f in paths)
        logging.info("[{}] contains {} samples {} steps".format(rank, n, n // args.batch_size))
        steps_cap = None          # per epoch, below: the loader re-shards the files every epoch
        loader = DataLoaderTrain(teacher_embs=teacher_embs, news_index=news_index, news_combined=news_combined, word_dict=None,
                                 data_dir=args.train_data_dir, filename_pat=args.filename_pat, args=args, world_size=size,
                                 worker_rank=rank, cuda_device_idx=local, enable_prefetch=True, enable_shuffle=True,
                                 enable_gpu=True, resident=args.resident_tables)
        dev_news, dev_tab = getattr(loader, "dev_news", None), loader.dev_tables
        batches = lambda: iter(loader)

        def epoch_cap():
            # uneven shards: every rank runs the step count the shortest shard of THIS epoch allows (the loader re-shards the
            # files with the epoch number as shuffle seed; the last batch may be short, streaming.py:76), otherwise the rank
            # that finishes first leaves the others waiting in the gradient all-reduce
            cap = dist.min_over_ranks(loader.next_epoch_batches(stat))
            if size > 1:
                logging.info("[{}] {} steps this epoch on every rank (shortest shard)".format(rank, cap))
            return cap

    if args.cache_frozen_layers and dev_news is not None and eng.build_frozen_cache(dev_news):
        logging.info("[%d] frozen layers 0..%d cached for %d news (%.2f GB)" % (
            rank, eng.lo - 1, dev_news.shape[0], eng.fcache[0].numel() * 4 / 1e9))
    logging.info("Training...")
    for ep in range(args.start_epoch, args.epochs):
        loss_sum, acc_sum, t0 = torch.zeros((), device="cuda"), torch.zeros((), device="cuda"), time.time()
        if not args.synthetic:
            steps_cap = epoch_cap()
        for cnt, batch in enumerate(batches()):
            if cnt > args.max_steps_per_epoch or cnt >= steps_cap:
                break
            if isinstance(batch, IndexBatch):
                h, m, c, y, plan = batch
                if plmnr:
                    total, y_student = model.forward_indexed(dev_news, h, m, c, y, plan)
                else:
                    total, distill, emb, target, y_student = model.forward_indexed(dev_news, h, m, c, y, dev_tab, plan)
            elif plmnr:
                h, m, c, y = batch[:4]
                total, y_student = model(h, m, c, y)
            else:
                h, m, c, y, th, tc = batch
                total, distill, emb, target, y_student = model(h, m, c, y, th, tc)
            loss_sum += total.detach()
            acc_sum += utils.acc(y, y_student)
            optimizer.zero_grad()
            total.backward()
            optimizer.step()
            if cnt % args.log_steps == 0:
                d = max(cnt, 1)
                sc = eng.scaler          # fp16: steps whose 16-bit backward overflowed are skipped on the device (engine.LossScaler)
                logging.info("[{}] Ed: {}, train_loss: {:.5f}, acc: {:.5f}, {:.1f} impressions/s{}".format(
                    rank, cnt * args.batch_size, loss_sum.item() / d, acc_sum.item() / d,
                    size * cnt * args.batch_size / max(time.time() - t0, 1e-9),
                    ", loss scale {:g}, {} steps skipped (fp16 overflow)".format(eng.gscale, sc.skipped) if sc.enabled and sc.skipped else ""))
        print(ep + 1)
        if eng.scaler.enabled:
            eng.scaler.drain(eng)        # the epoch's last two overflow answers, before the step count goes into a log line or a file
        if rank == 0:                                                        # run.py:205-214
            os.makedirs(args.model_dir, exist_ok=True)
            ckpt_path = os.path.join(args.model_dir, f"epoch-{ep + 1}.pt")
            torch.save({"model_state_dict": {k: v.cpu() for k, v in model.state_dict().items()},
                        "category_dict": category_dict, "word_dict": None, "subcategory_dict": subcategory_dict}, ckpt_path)
            logging.info(f"Model saved to {ckpt_path}")
    if not args.synthetic:
        loader.join()


def _news_table(args, data_dir, mode):
    from preprocess import get_doc_input_bert, read_news_bert
    if mode == "train":
        news, news_index, cat, sub = read_news_bert(os.path.join(data_dir, "news.tsv"), args, mode="train")
    else:
        news, news_index = read_news_bert(os.path.join(data_dir, "news.tsv"), args, mode="test")
        cat, sub = {}, {}
    title, mask, _, _ = get_doc_input_bert(news, news_index, cat, sub, args)
    return news_index, np.concatenate([title, mask], axis=-1).astype(np.int32)


def _forward_engine(args, n_layers, sd, add_prefix=""):
    """Frozen engine (no teachers, nothing trainable).  sd: reference Model keys ("student. ...") or, with
    add_prefix="student.", PLM-NR ModelBert keys ("news_encoder. ...", "user_encoder. ...")."""
    import types
    import engine as E
    from model_bert import engine_config_from_args
    a = types.SimpleNamespace(**vars(args))
    a.num_student_layers, a.bert_trainable_layer = n_layers, []
    cfg = engine_config_from_args(a, num_teachers=0)
    eng = E.Engine(cfg, "cuda:%d" % torch.cuda.current_device(), max_batch=args.batch_size, dtype=getattr(args, "dtype", "fp16"))
    src = {add_prefix + k: v for k, v in sd.items()}
    eng.load_state_dict({k: src[k] for k in eng.shapes})
    return eng


def get_teacher_emb(args):
    """run.py:382-460: every train news through each teacher's news encoder -> pickled (n+1, D) float32 tables."""
    utils.init_hvd_cuda(False, args.enable_gpu)
    _, news_combined = _news_table(args, args.train_data_dir, "train")
    dev_news = torch.from_numpy(news_combined).cuda()
    for ckpt_path, out_path in zip(args.teacher_ckpts, args.teacher_emb_paths):
        sd = torch.load(ckpt_path, map_location="cpu")["model_state_dict"]          # PLM-NR keys: news_encoder.* / user_encoder.*
        eng = _forward_engine(args, args.num_hidden_layers, sd, add_prefix="student.")
        logging.info(f"loaded teacher model: {ckpt_path}")
        news_scoring = eng.encode_news(dev_news).cpu().numpy()
        logging.info("news scoring num: {}".format(news_scoring.shape[0]))
        with open(out_path, "wb") as f:
            pickle.dump(news_scoring, f)
        logging.info(f"teacher embedding saved at {out_path}")
        del eng
        torch.cuda.empty_cache()


def test(args, collect=None):
    """run.py:219-379: encode all test news once, user vectors per impression batch, host metrics, scalar all-reduce.
    collect: optional list that receives, per impression in file order, (scores, labels) - the reference keeps only the sums;
    the quality tests compare the per-impression rankings with the reference's."""
    import dist
    from dataloader import DataLoaderTest
    from metrics import mrr_score, ndcg_score, roc_auc_score
    size, rank, local = utils.init_hvd_cuda(args.enable_hvd, args.enable_gpu)
    ckpt_path = utils.get_checkpoint(args.model_dir, args.load_ckpt_name) if args.load_ckpt_name else utils.latest_checkpoint(args.model_dir)
    assert ckpt_path is not None, "No ckpt found"
    sd = torch.load(ckpt_path, map_location="cpu")["model_state_dict"]
    eng = _forward_engine(args, args.num_student_layers, sd)
    logging.info(f"Model loaded from {ckpt_path}")
    news_index, news_combined = _news_table(args, args.test_data_dir, "test")
    news_scoring = eng.encode_news(torch.from_numpy(news_combined).cuda())
    scoring_host = news_scoring.cpu().numpy()
    logging.info("news scoring num: {}".format(scoring_host.shape[0]))
    loader = DataLoaderTest(news_index=news_index, data_dir=args.test_data_dir, filename_pat=args.filename_pat, args=args,
                            world_size=size, worker_rank=rank, cuda_device_idx=local, enable_prefetch=True,
                            enable_shuffle=False, enable_gpu=True)
    sums, n_local = np.zeros(4), 0
    cnt_metric = 0
    for cnt, (h, m, cands, labels) in enumerate(loader):
        n_local += h.shape[0]
        users = eng.user_vectors(news_scoring, h, m).cpu().numpy()
        for u, c, y in zip(users, cands, labels):
            if collect is not None:
                collect.append((scoring_host[c] @ u, y))
            if y.mean() == 0 or y.mean() == 1:
                continue
            score = scoring_host[c] @ u
            sums += [roc_auc_score(y, score), mrr_score(y, score), ndcg_score(y, score, 5), ndcg_score(y, score, 10)]
            cnt_metric += 1
        if cnt % args.log_steps == 0 and cnt_metric:
            logging.info("[{}] Ed: {}: {}".format(rank, n_local, "\t".join("{:0.2f}".format(x * 100) for x in sums / cnt_metric)))
    loader.join()
    logging.info("[{}] local_sample_num: {}".format(rank, n_local))
    tot = torch.tensor([float(n_local)] + list(sums), dtype=torch.float64, device="cuda")
    if size > 1:
        torch.distributed.all_reduce(tot)                                               # hvd.allreduce(Sum), run.py:372-376
    if rank == 0:
        t = tot.cpu().numpy()
        # the reference divides the metric sums by the SAMPLE count (run.py:377-379), impressions skipped above included
        logging.info("[{}] Ed: {}: {}".format(rank, int(t[0]), "\t".join("{:0.2f}".format(x * 100) for x in t[1:] / t[0])))
    return sums, n_local, cnt_metric


if __name__ == "__main__":
    utils.setuplogger()
    args = parse_args()
    {"train": train, "test": test, "get_teacher_emb": get_teacher_emb}[args.mode](args)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
