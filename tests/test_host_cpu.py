"""CPU: host side of the boundary -- flag surface, state_dict schema, file sharding, sample decoding (bit-exact
against vectors captured from the reference), TF-free line streaming, and the data-parallel gradient
sync on a 2-rank gloo group."""
import json
import os
import random
import types

import numpy as np
import pytest
import torch

from helpers import GOLDEN

IFACE = json.load(open(os.path.join(GOLDEN, "interface.json")))
DATA = os.path.join(GOLDEN, "data")


def test_every_reference_flag_with_its_default():
    import parameters
    a = vars(parameters.parse_args([]))
    for k, v in IFACE["flags"].items():
        assert k in a, "flag --%s missing" % k
        assert a[k] == v, "--%s default %r != reference %r" % (k, a[k], v)
    b = parameters.parse_args("--mode train --bert_trainable_layer 2 3 --user_log_mask False --coef 0.2 "
                              "--teacher_ckpts a b --num_teachers 2".split())
    assert b.bert_trainable_layer == [2, 3] and b.user_log_mask is False and b.teacher_ckpts == ["a", "b"]
    with pytest.raises(SystemExit):
        parameters.parse_args(["--mode", "bogus"])


def test_state_dict_schema_and_trainable_set_match_reference():
    import engine as E
    cfg = E.EngineConfig(n_layers=4, trainable_layers=(2, 3), num_teachers=4)
    shapes = E.param_shapes(cfg)
    assert {k: list(v) for k, v in shapes.items()} == IFACE["state_dict"]
    assert sorted(k for k in shapes if E.is_trainable(cfg, k)) == sorted(IFACE["trainable_4layer_23"])
    eng = E.Engine(cfg, device="cpu", max_batch=1)        # storage layout only; no kernel is launched
    assert sum(v.numel() for v in eng.params.values()) == 53654504
    assert sum(v.numel() for k, v in eng.params.items() if E.is_trainable(cfg, k)) == 14841634
    # parameters are views of the flat buffers, 16-byte aligned, non-overlapping
    spans = sorted((eng.slot[k][1], eng.slot[k][1] + eng.slot[k][2]) for k in shapes if eng.slot[k][0])
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    assert all(v.data_ptr() % 16 == 0 for k, v in eng.params.items() if v.numel() >= 4)
    # gradient buckets tile the trainable buffer exactly once
    br = sorted(eng.bucket_ranges())
    assert br[0][0] == 0 and br[-1][1] == eng.n_train and all(a[1] == b[0] for a, b in zip(br, br[1:]))
    # stacked groups the kernels read as one array
    q, k_ = (E.layer_param_order(3)[0], E.layer_param_order(3)[1])
    assert eng.off(k_) == eng.off(q) + 768 * 768
    assert eng.off("teachers.1.attn.att_fc1.weight") == eng.off("teachers.0.attn.att_fc1.weight") + 200 * 256


def test_worker_file_sharding_matches_reference():
    import streaming
    z = np.load(os.path.join(GOLDEN, "datapath.npz"))
    for w in (1, 2, 3):
        seen = []
        for r in range(w):
            for sh in (0, 1):
                got = [os.path.basename(x) for x in streaming.get_worker_files(DATA, r, w, "behaviors_np4_*.tsv", bool(sh), 3)]
                assert got == [str(x) for x in z["files_w%d_r%d_s%d" % (w, r, sh)]]
            seen += [os.path.basename(x) for x in streaming.get_worker_files(DATA, r, w, "behaviors_np4_*.tsv")]
        assert sorted(seen) == ["behaviors_np4_%d.tsv" % i for i in range(3)]       # disjoint and complete
    stat = streaming.get_stat(DATA, "behaviors_np4_*.tsv")
    assert sorted(stat.values()) == [4, 4, 4]


def _loader(resident=False, batch_size=5, world=1, rank=0, shuffle=False):
    import dataloader
    import hashinit
    z = np.load(os.path.join(GOLDEN, "datapath.npz"))
    news_index = {str(k): int(v) for k, v in zip(z["news_ids"], z["news_index"])}
    comb = z["news_combined"]
    temb = [hashinit.hash_normal(11, "dp_temb%d" % i, (comb.shape[0], 8)) for i in range(2)]
    args = types.SimpleNamespace(npratio=4, user_log_length=50, batch_size=batch_size, shuffle_buffer_size=7, num_teachers=2)
    dl = dataloader.DataLoaderTrain(DATA, "behaviors_np4_*.tsv", args, world, rank, 0, news_index, comb, temb,
                                    enable_prefetch=True, enable_shuffle=shuffle, enable_gpu=False, resident=resident)
    return dl, z


def test_sample_decoding_is_bit_exact_with_reference():
    dl, z = _loader()
    lines = [str(x).encode() for x in z["lines"]]
    random.seed(7)
    out = dl._process(lines)
    assert out[0].dtype == torch.int64 and np.array_equal(out[0].numpy(), z["log_ids"])
    assert out[1].dtype == torch.float32 and np.array_equal(out[1].numpy(), z["log_mask"])
    assert np.array_equal(out[2].numpy(), z["input_ids"]) and np.array_equal(out[3].numpy(), z["targets"])
    assert np.array_equal(out[4][0].numpy(), z["th0"]) and np.array_equal(out[5][1].numpy(), z["tc1"])
    # edge cases of the golden lines: empty history, >50 clicks, unknown ids
    h, m, c, y = dl.decode([b"9\tU\tt\t\tN1\tN2 N3 N4 N5"])
    assert h.sum() == 0 and m.sum() == 0


def test_stream_covers_every_line_once_with_short_last_batch():
    dl, z = _loader(batch_size=5)
    got = []
    sizes = []
    for b in dl:
        sizes.append(b[0].shape[0])
        got.append(b[3])
    assert sizes == [5, 5, 2] and sum(sizes) == len(z["lines"])
    # second epoch restarts cleanly; shuffled epoch still covers all lines
    assert sum(b[0].shape[0] for b in dl) == 12
    dl2, _ = _loader(batch_size=4, shuffle=True)
    assert sum(b[0].shape[0] for b in dl2) == 12
    dl2.join()
    # two workers see disjoint files: 2 + 1 files -> 8 + 4 lines
    n = [sum(b[0].shape[0] for b in _loader(world=2, rank=r)[0]) for r in range(2)]
    assert sorted(n) == [4, 8]


def test_decode_process_yields_the_same_batches_as_the_producer_thread():
    """run.py --decode_process (resident mode): the epoch's sampler + decode + de-duplication plan run in a spawned child
    process.  Same batches, same label draws (the child continues this process's global `random` stream and hands the state
    back), same plans as the in-thread producer - here with the device hand-over replaced by a pass-through (no GPU)."""
    import dataloader

    def collect(use_process):
        dl, _ = _loader(batch_size=5)                 # no shuffle buffer: its generator is seeded from the OS, as tf.data's is
        dl.resident, dl.enable_gpu, dl.dedup, dl.decode_process = True, True, True, use_process
        dl._to_device = lambda h, m, c, y, plan: (np.asarray(h).copy(), np.asarray(m).copy(), np.asarray(c).copy(), np.asarray(y).copy(),
                                                  None if plan is None else (plan.uniq.copy(), plan.inv.copy(), plan.order.copy(),
                                                                             plan.seg.copy(), plan.n_enc, plan.n_unique, plan.n_slots))
        import torch as _t
        _set = _t.cuda.set_device
        _t.cuda.set_device = lambda *_: None          # the producer thread's set_device (dataloader.py:86-88) needs no GPU here
        try:
            random.seed(123)
            out = [list(dl) for _ in range(2)]            # two epochs: the state handed back feeds the second one
            dl.join()
        finally:
            _t.cuda.set_device = _set
        return out, random.getstate()

    (a, sa), (b, sb) = collect(False), collect(True)
    assert sa == sb
    for ea, eb in zip(a, b):
        assert len(ea) == len(eb) == 3
        for x, y in zip(ea, eb):
            for u, v in zip(x[:4], y[:4]):
                assert np.array_equal(u, v)
            assert (x[4] is None) == (y[4] is None)
            if x[4] is not None:
                assert all(np.array_equal(u, v) for u, v in zip(x[4][:4], y[4][:4])) and x[4][4:] == y[4][4:]


def test_news_table_layout_with_stub_tokenizer():
    import preprocess
    L = 6

    def tok(text, max_length, padding, truncation):
        ids = [101] + [1000 + len(w) for w in text.split()][:max_length - 2] + [102]
        am = [1] * len(ids) + [0] * (max_length - len(ids))
        return {"input_ids": ids + [0] * (max_length - len(ids)), "attention_mask": am}

    args = types.SimpleNamespace(num_words_title=L, tokenizer_name=None)
    news, idx, cat, sub = preprocess.read_news_bert(os.path.join(DATA, "news.tsv"), args, "train", tokenizer=tok)
    t, m, _, _ = preprocess.get_doc_input_bert(news, idx, cat, sub, args)
    assert list(idx.values()) == list(range(1, 41)) and t.shape == (41, L) and t.dtype == np.int32
    assert (t[0] == 0).all() and (m[0] == 0).all() and (t[1:, 0] == 101).all()


def _sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import dist
    w, r, _ = dist.init("gloo")
    flat = torch.arange(1000, dtype=torch.float32) * (r + 1)
    ranges = [(600, 1000), (300, 600), (0, 300)]
    gs = dist.GradSync(flat, ranges, w)
    for b in range(3):
        gs.launch(b)
    # per-bucket completion (Engine.step runs each bucket's optimiser slice behind its own all-reduce only): bucket 0 is
    # final after wait_bucket(0) whatever the later ones are doing; waiting twice / for an unlaunched bucket is a no-op
    assert sorted(gs.pending) == [0, 1, 2]
    gs.wait_bucket(0)
    assert torch.equal(flat[600:1000], torch.arange(600, 1000, dtype=torch.float32) * 3) and sorted(gs.pending) == [1, 2]
    gs.wait_bucket(0)
    gs.wait()
    assert not gs.pending
    # one collective per trainable layer (merge_layers): buckets [heads | FFN, attention of layer 1 | FFN, attention of layer 0];
    # a layer's collective starts when BOTH its buckets have been handed over, and waiting for either waits for it
    flat2 = torch.arange(1000, dtype=torch.float32) * (r + 1)
    g2 = dist.GradSync(flat2, [(900, 1000), (700, 900), (500, 700), (250, 500), (0, 250)], w, merge_layers=True, algo="rs_ag")
    assert g2.groups == [[0], [1, 2], [3, 4]] and g2.group_range == [(900, 1000), (500, 900), (0, 500)] and g2.algo == "allreduce"   # gloo: no rs_ag
    g2.launch(0); g2.launch(1)
    assert sorted(g2.pending) == [0]
    g2.launch(2); g2.launch(3); g2.launch(4)
    assert sorted(g2.pending) == [0, 1, 2]
    g2.wait_bucket(2)
    assert torch.equal(flat2[500:900], torch.arange(500, 900, dtype=torch.float32) * 3) and sorted(g2.pending) == [0, 2]
    g2.wait_bucket(1)
    g2.wait()
    assert not g2.pending and torch.equal(flat2, torch.arange(1000, dtype=torch.float32) * 3)
    p = torch.full((10,), float(r))
    dist.broadcast_flat([p])
    dist.barrier()
    q.put((r, flat.numpy().copy(), gs.scale, p.numpy().copy()))      # by value: the child exits right after


def test_gradient_average_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    ps = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    want = torch.arange(1000, dtype=torch.float32) * 3          # rank0 * 1 + rank1 * 2
    for r, flat, scale, p in res:
        assert np.array_equal(flat, want.numpy()) and scale == 0.5   # sum on the wire, 1/world folded into AMSGrad
        assert np.array_equal(p, np.zeros(10, np.float32))            # parameters broadcast from rank 0


def _uneven_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import dist
    dist.init("gloo")
    n_lines, B = (1003, 996)[rank], 32                      # uneven shards: 32 vs 32 batches ... and 1003 -> 32, 996 -> 32
    cap = dist.min_over_ranks(-(-n_lines // B))
    cap2 = dist.min_over_ranks((7, 5)[rank])
    # every rank runs `cap` steps with one collective each: no rank is left waiting (the reference would hang, run.py:176)
    t = torch.zeros(1)
    for _ in range(cap2):
        torch.distributed.all_reduce(t)
    dist.barrier()
    q.put((rank, cap, cap2))


def test_uneven_shards_agree_on_the_shortest_step_count_gloo():
    """SURVEY.md section 5 (failure detection): ranks read sorted(files)[r::W], so batch counts differ; run.py caps every
    rank's epoch at the minimum over ranks (dist.min_over_ranks) instead of letting the gradient all-reduce hang."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    ps = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert [r[1] for r in res] == [32, 32] and [r[2] for r in res] == [5, 5]


def test_metrics_match_sklearn_and_reference_formulas():
    import metrics
    from sklearn.metrics import roc_auc_score
    rs = np.random.RandomState(0)
    for _ in range(100):
        n = rs.randint(2, 30)
        y = rs.randint(0, 2, n)
        if y.mean() in (0, 1):
            continue
        s = np.round(rs.randn(n), 1)           # ties included
        assert abs(metrics.roc_auc_score(y, s) - roc_auc_score(y, s)) < 1e-12
    y, s = np.array([0, 1, 0, 1]), np.array([0.1, 0.9, 0.8, 0.3])
    assert abs(metrics.mrr_score(y, s) - (1 / 1 + 1 / 3) / 2) < 1e-12
    assert abs(metrics.ndcg_score(y, s, 10) - (1 + 1 / np.log2(4)) / (1 + 1 / np.log2(3))) < 1e-12


def test_eval_loader_index_level():
    import dataloader
    z = np.load(os.path.join(GOLDEN, "datapath.npz"))
    news_index = {str(k): int(v) for k, v in zip(z["news_ids"], z["news_index"])}
    args = types.SimpleNamespace(npratio=4, user_log_length=50, batch_size=2, shuffle_buffer_size=7, num_teachers=0)
    dl = dataloader.DataLoaderTest(DATA, "none", args, 1, 0, 0, news_index, enable_gpu=False)
    h, m, c, y = dl._process([b"1\tU1\tt\tN1 N2 N999\tN3-1 N4-0 N777-0"])
    assert h.shape == (1, 50) and h[0, -3:].tolist() == [news_index["N1"], news_index["N2"], 0] and m[0].sum() == 3
    assert c[0].tolist() == [news_index["N3"], news_index["N4"], 0] and y[0].tolist() == [1, 0, 0]


def test_split_file_writes_loader_format(tmp_path):
    import split_file
    import streaming
    src = tmp_path / "behaviors.tsv"
    src.write_text("1\tU1\tt1\tN1 N2\tN3-1 N4-0 N5-0 N6-0 N7-0 N8-1\n"
                   "2\tU2\tt2\t\tN3-0 N4-0\n"                 # no positive: dropped
                   "3\tU3\tt3\tN9\tN1-1 N2-0\n")              # 1 negative, 4 wanted: sampled from the repeated pool
    paths, n = split_file.split(str(src), 2, npratio=4, seed=0)
    assert n == 3 and [os.path.basename(p) for p in paths] == ["behaviors_np4_0.tsv", "behaviors_np4_1.tsv"]
    lines = [l for p in paths for l in open(p).read().splitlines()]
    assert len(lines) == 3
    for l in lines:
        f = l.split("\t")
        assert len(f) == 6 and len(f[4].split()) == 1 and len(f[5].split()) == 4
    assert sorted(l.split("\t")[4] for l in lines) == ["N1", "N3", "N8"]
    assert sorted(streaming.get_stat(str(tmp_path), "behaviors_np4_*.tsv").values()) == [1, 2]


def test_pretrained_weight_import_matches_reference():
    """tnlrv3/convert_state_dict.py:39-71 + the position-table resize of tnlrv3/modeling.py:90-118
    against tensors captured from the reference's own functions (tests/golden/convert.npz)."""
    import torch
    from helpers import unilm_checkpoint
    from tnlrv3 import convert_state_dict as C
    z = np.load(os.path.join(GOLDEN, "convert.npz"))
    seed, H, nl, A, I, vocab, max_pos = [int(x) for x in z["dims"]]
    raw = unilm_checkpoint(seed, H, nl, A, I, vocab, max_pos)
    conv = C.load_model(dict(raw))
    assert sorted(conv) == [str(k) for k in z["keys"]]
    for k, v in conv.items():
        ref = z["v." + k]
        assert tuple(v.shape) == ref.shape and np.array_equal(v.numpy(), ref), k
    assert C.state_dict_convert["tnlrv3"] is C.load_model
    pos = conv[C.POS_KEY]
    g = C.resize_position_embeddings(pos, 16, 0.02, None)
    ref = z["pos.grow"]
    assert np.array_equal(g[:max_pos].numpy(), ref[:max_pos])                     # old rows kept
    assert not np.array_equal(g[max_pos:].numpy(), np.zeros_like(ref[max_pos:]))  # new rows drawn N(0, 0.02)
    assert abs(float(g[max_pos:].std()) - 0.02) < 0.006 and abs(float(ref[max_pos:].std()) - 0.02) < 0.006
    assert np.array_equal(C.resize_position_embeddings(pos, 16, 0.02, True).numpy(), z["pos.grow_reuse"])   # tiled: exact
    assert np.array_equal(C.resize_position_embeddings(pos, 4, 0.02, None).numpy(), z["pos.shrink"])
    assert np.array_equal(C.resize_position_embeddings(pos, max_pos, 0.02, None).numpy(), z["pos.same"])

    # what from_pretrained leaves in a 2-layer student: converted keys under the module prefix, extra layer dropped
    wanted = {"student.news_encoder.bert_model." + k: tuple(v.shape) for k, v in conv.items()
              if not k.startswith("bert.encoder.layer.2.") and not k.startswith("cls.")}
    wanted["student.news_encoder.bert_model.bert.embeddings.position_embeddings.weight"] = (12, H)
    wanted["student.news_encoder.bert_model.classifier.weight"] = (2, H)
    state, missing, unexpected = C.student_state_from_pretrained(raw, wanted, 2, 12, reuse_position_embedding=True)
    assert missing == ["student.news_encoder.bert_model.classifier.weight"]
    assert "cls.predictions.bias" in unexpected and any(k.startswith("bert.encoder.layer.2.") for k in unexpected)
    assert set(state) == set(wanted) - set(missing)
    assert state["student.news_encoder.bert_model.bert.embeddings.position_embeddings.weight"].shape == (12, H)
    with pytest.raises(ValueError):
        C.student_state_from_pretrained(raw, dict(wanted, **{"student.news_encoder.bert_model.bert.pooler.dense.bias": (H + 1,)}), 2, 12)


def test_dedup_plan_is_a_bijection_onto_distinct_news():
    """dedup.build_plan: integer-exact grouping of the B*(U+C) news slots of a batch by distinct news id."""
    import synth
    from dedup import build_plan
    h, m, c, y = synth.impressions(9, 8, 300, 50, 5)
    p = build_plan(h, c, quantum=64)
    slots = np.concatenate([h.ravel(), c.ravel()])
    assert p.n_slots == slots.size and p.n_unique == np.unique(slots).size and p.n_enc % 64 == 0 and p.n_enc >= p.n_unique
    assert np.array_equal(p.uniq[p.inv], slots)                              # expansion reproduces every slot
    assert (p.uniq[p.n_unique:] == 0).all()                                  # padding rows are the pad news
    assert np.array_equal(np.sort(p.order), np.arange(slots.size))           # every slot in exactly one group
    for u in range(p.n_enc):
        grp = p.order[p.seg[u]:p.seg[u + 1]]
        assert (p.inv[grp] == u).all() and (np.diff(grp) > 0).all()          # stable: fixed summation order
    assert p.seg[p.n_unique] == slots.size and (p.seg[p.n_unique:] == slots.size).all()   # padding rows own no slot
    # all-distinct batch: nothing to save -> no plan
    assert build_plan(np.arange(1, 101).reshape(2, 50), np.arange(101, 111).reshape(2, 5)) is None
    # empty histories (all pad) collapse to one row
    q = build_plan(np.zeros((2, 50), np.int64), np.array([[1, 2, 3, 4, 5], [1, 2, 3, 4, 6]]))
    assert q.n_unique == 7 and q.seg[1] == 100


@pytest.mark.parametrize("pooling,heads", [("cls", 0), ("mean", 16), ("att", 16)])
def test_state_dict_schema_of_pooling_and_nrms_variants(pooling, heads):
    """the engine's parameter schema for args.pooling in {cls, mean} / args.model == 'NRMS' equals the
    key set the reference instantiates (helpers.state_shapes is what the golden harness loaded into the reference)."""
    import engine as E
    from helpers import FULL, state_shapes
    cfg = E.EngineConfig(n_layers=2, trainable_layers=(0, 1), num_teachers=2, pooling=pooling, nrms_heads=heads)
    want = state_shapes(FULL, 2, 256, 2, pooling, heads)
    got = E.param_shapes(cfg)
    assert {k: tuple(v) for k, v in got.items()} == {k: tuple(v) for k, v in want.items()}
    eng = E.Engine(cfg, device="cpu", max_batch=1)
    assert all(E.is_trainable(cfg, k) for k in got if "multi_head_self_attn" in k and k.startswith("student."))
    if heads:   # [W_Q; W_K; W_V] of every encoder are one contiguous (3D, D) block, teachers back to back
        q, k_ = "student.user_encoder.multi_head_self_attn.W_Q.weight", "student.user_encoder.multi_head_self_attn.W_K.weight"
        assert eng.off(k_) == eng.off(q) + 256 * 256
        assert eng.off("teachers.1.multi_head_self_attn.W_Q.weight") == eng.off("teachers.0.multi_head_self_attn.W_Q.weight") + 3 * 256 * 256
    br = sorted(eng.bucket_ranges())
    assert br[0][0] == 0 and br[-1][1] == eng.n_train and all(a[1] == b[0] for a, b in zip(br, br[1:]))


def _epoch_cap_worker(rank, world, port, data_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import dist
    from dataloader import DataLoaderTrain
    from streaming import get_stat
    dist.init("gloo")
    args = types.SimpleNamespace(npratio=4, user_log_length=5, batch_size=8, shuffle_buffer_size=16, num_teachers=0, dedup_news=False)
    news_index = {"N%d" % i: i for i in range(1, 30)}
    loader = DataLoaderTrain(data_dir=data_dir, filename_pat="behaviors_*.tsv", args=args, world_size=world, worker_rank=rank,
                             cuda_device_idx=0, news_index=news_index, news_combined=np.zeros((30, 8), np.int32), teacher_embs=[],
                             enable_prefetch=True, enable_shuffle=True, enable_gpu=False, resident=False)
    stat = get_stat(data_dir, "behaviors_*.tsv")
    out = []
    for ep in range(3):
        mine = loader.next_epoch_batches(stat)               # what THIS epoch's re-sharded file set gives this worker
        cap = dist.min_over_ranks(mine)                      # run.py: before every epoch
        got = steps = 0
        t = torch.zeros(1)
        for batch in loader:                                 # the loop of run.py: one collective per step, `cap` steps on every rank
            got += 1
            if steps < cap:
                torch.distributed.all_reduce(t)
                steps += 1
        out.append((mine, cap, got, steps))
    loader.join()
    dist.barrier()
    q.put((rank, out))


def test_step_cap_follows_the_per_epoch_resharding_gloo(tmp_path):
    """The loader re-shards the files EVERY epoch (shuffle seed = epoch number, dataloader.py:61-70), so with unequal files a
    worker's batch count changes from epoch to epoch: the cap is recomputed per epoch from the loader's own arguments
    (DataLoaderTrain.next_epoch_batches + dist.min_over_ranks) -- a cap taken once from the epoch-0 file set would let a rank
    run dry in a later epoch while the other waits in the all-reduce."""
    import torch.multiprocessing as mp
    from streaming import get_worker_files, shard_files
    sizes = {"behaviors_0.tsv": 100, "behaviors_1.tsv": 37, "behaviors_2.tsv": 71, "behaviors_3.tsv": 9, "behaviors_4.tsv": 55}
    for name, n in sizes.items():
        with open(tmp_path / name, "w") as f:
            for i in range(n):
                f.write("%d\tU%d\tt\tN1 N2 N3\tN4\tN5 N6 N7 N8\n" % (i, i))
    d = str(tmp_path)
    for seed in range(6):                                    # the side-effect-free shard rule == the reference's (which seeds `random`)
        for r in range(2):
            st = random.getstate()
            assert shard_files(d, r, 2, "behaviors_*.tsv", True, seed) == get_worker_files(d, r, 2, "behaviors_*.tsv", True, seed)
            random.setstate(st)
            assert shard_files(d, r, 2, "behaviors_*.tsv", False, seed) == get_worker_files(d, r, 2, "behaviors_*.tsv", False, seed)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    ps = [ctx.Process(target=_epoch_cap_worker, args=(r, 2, port, d, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
    caps_seen = set()
    for ep in range(3):
        (m0, c0, g0, s0), (m1, c1, g1, s1) = res[0][ep], res[1][ep]
        assert g0 == m0 and g1 == m1                         # the prediction is what the iterator really yields
        assert c0 == c1 == min(m0, m1) and s0 == s1 == c0    # both ranks ran the same number of collectives
        files = [shard_files(d, r, 2, "behaviors_*.tsv", True, ep) for r in range(2)]
        assert [m0, m1] == [-(-sum(sizes[os.path.basename(f)] for f in fs) // 8) for fs in files]
        caps_seen.add((m0, m1))
    assert len(caps_seen) > 1                                # the file sets really changed between epochs


def test_decode_worker_never_imports_torch():
    """The spawned decoder of the resident feed must come up without torch (and must not touch a GPU): importing its module, and
    decoding a batch in it, works in an interpreter where torch cannot be imported."""
    import subprocess, sys
    code = ("import sys; sys.modules['torch'] = None; sys.path.insert(0, %r)\n"
            "import decode_worker as d\n"
            "h, m, c, y = d.decode_lines([b'0\\tu\\tt\\tN1 N2 N9\\tN3\\tN4 N5'], {'N1': 1, 'N2': 2, 'N3': 3, 'N4': 4, 'N5': 5}, 4, 2)\n"
            "assert h.tolist() == [[0, 1, 2, 0]] and m.tolist() == [[0.0, 1.0, 1.0, 1.0]], (h, m)\n"
            "assert sorted(c[0].tolist()) == [3, 4, 5] and c[0, y[0]] == 3, (c, y)\n"
            "assert 'torch' not in [k for k, v in sys.modules.items() if v is not None]\n") % os.path.join(os.path.dirname(GOLDEN), "..", "tiny-newsrec_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


def test_spawned_decoder_does_not_reimport_main(tmp_path):
    """multiprocessing's "spawn" re-imports the parent's __main__ in every child - for run.py that is `import torch` at the start
    of every epoch, in the process that exists to stay away from it.  decode_worker.start_without_main starts the child without:
    spawned from a SCRIPT that has imported torch, the child has neither torch nor the script's module."""
    import subprocess, sys
    script = tmp_path / "parent_main.py"
    script.write_text(
        "import sys, multiprocessing as mp\n"
        "sys.path.insert(0, %r)\n"
        "import torch\n"
        "MARK = 'parent main imported'\n"
        "import decode_worker as d\n"
        "if __name__ == '__main__':\n"
        "    ctx = mp.get_context('spawn')\n"
        "    out = {}\n"
        "    for name, start in (('plain', lambda p: p.start()), ('hidden', d.start_without_main)):\n"
        "        q = ctx.Queue()\n"
        "        p = ctx.Process(target=d.report_modules, args=(q, ['torch', '__mp_main__']), daemon=True)\n"
        "        start(p); out[name] = q.get(timeout=100); p.join(20)\n"
        "    assert 'torch' in out['plain'] and '__mp_main__' in out['plain'], out      # what multiprocessing does by itself\n"
        "    assert 'torch' not in out['hidden'], out\n"
        "    assert sys.modules['__main__'].__file__.endswith('parent_main.py')          # restored\n"
        % os.path.join(os.path.dirname(GOLDEN), "..", "tiny-newsrec_amd"))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr


def test_loss_scaler_host_logic():
    """engine.LossScaler.poll (host half of the fp16 loss scale; the reference trains in fp32, run.py:134,194-195): an answer is
    used exactly two steps after its step, an overflow halves the multiplier and takes the skipped step out of Adam's count,
    `growth_interval` clean answers in a row double it (up to max_mult), 16 skips in a row warn once about a forward overflow."""
    import logging
    import engine as E

    class Ev:
        def synchronize(self):
            pass
    sc = E.LossScaler("cpu", True, growth_interval=3, max_mult=4.0)
    assert not sc.enabled                      # no device: nothing is launched, poll is pure host logic
    eng = types.SimpleNamespace(step_count=0, gscale=1024.0)
    overflow_at = {4, 5, 12}
    seen = []
    for step in range(1, 16):
        used = sc.mult                         # Engine.step: the multiplier the backward ran with, poll, stamp += 1, launches, record
        sc.poll(eng)
        sc.stamp += 1
        eng.step_count += 1
        sc.pending.append((sc.stamp, [sc.stamp if sc.stamp in overflow_at else 0], Ev(), used))
        seen.append((sc.mult, sc.skipped, eng.step_count))
    # the answer of step s is read at the start of step s + 2; an overflow takes the multiplier to half of what THAT step ran with
    # (steps 4 and 5 both ran at 1: one halving between them; the growth to 2 in between does not survive)
    assert [m for m, _, _ in seen] == [1, 1, 1, 1, 2, 0.5, 0.5, 0.5, 0.5, 1, 1, 1, 2, 0.5, 0.5], seen
    assert [k for _, k, _ in seen] == [0, 0, 0, 0, 0, 1, 2, 2, 2, 2, 2, 2, 2, 3, 3]
    assert seen[-1][2] == 15 - 3               # three skipped steps taken out of Adam's count
    sc.drain(eng)
    assert not sc.pending
    # everything overflows (a forward overflow): one halving per three skipped steps (the two steps behind an overflow ran at
    # the same multiplier), the multiplier stops at min_mult, sixteen skips in a row warn once
    sc2 = E.LossScaler("cpu", True, min_mult=2.0 ** -4)
    eng2 = types.SimpleNamespace(step_count=100, gscale=1.0)
    records = []
    h = logging.Handler()
    h.emit = records.append
    logging.getLogger().addHandler(h)
    try:
        for s in range(1, 40):
            used = sc2.mult
            sc2.poll(eng2)
            sc2.stamp += 1
            sc2.pending.append((sc2.stamp, [sc2.stamp], Ev(), used))
            if s == 9:
                assert sc2.mult == 0.125 and sc2.skipped == 7     # answers 1 .. 7: halved by answers 1, 4 and 7
        sc2.drain(eng2)
    finally:
        logging.getLogger().removeHandler(h)
    assert sc2.skipped == 39 and sc2.mult == 2.0 ** -4 and eng2.step_count == 100 - 39
    assert sum("bf16" in r.getMessage() for r in records) == 1
    # the advisor's case (round 4): the scale has grown once, the next backward overflows BECAUSE of it; its answer comes two
    # steps late, so the two steps behind it overflow at the same scale: three skipped steps, ONE halving (back to where it was)
    sc3 = E.LossScaler("cpu", True, growth_interval=2, max_mult=64.0)
    eng3 = types.SimpleNamespace(step_count=0, gscale=1024.0)
    mults = []
    for step in range(1, 12):
        used = sc3.mult
        sc3.poll(eng3)
        sc3.stamp += 1
        eng3.step_count += 1
        bad = used > 1.0                       # this model overflows whenever its backward runs above the base scale
        sc3.pending.append((sc3.stamp, [sc3.stamp if bad else 0], Ev(), used))
        mults.append(sc3.mult)
    # answers 1, 2 clean -> x2 at the start of step 4; steps 5 and 6 run their backward at 2 and overflow, answers 3, 4 are still
    # clean -> x2 again at step 6, step 7 runs at 4; step 5's answer arrives at step 7 -> half of the 2 it ran with; the answers of
    # steps 6 and 7 (2 and 4) change nothing any more: three skipped steps, the multiplier back where it last worked
    assert mults == [1, 1, 1, 2, 2, 4, 1, 1, 1, 1, 2], mults
    assert sc3.skipped == 3 and min(mults) == 1
    sc3.reset(eng3)
    assert sc3.mult == 1.0 and not sc3.pending
