"""GPU: the reference's Python surface on the HIP engine -- model_bert.Model(args).forward(...) 5-tuple,
total_loss.backward() filling .grad, optimizer.step(), state_dict schema, and `python run.py --mode train`."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import GOLDEN, load_case   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IFACE = json.load(open(os.path.join(GOLDEN, "interface.json")))


def _args(z, cfg, T_):
    seed, B, _, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
    return types.SimpleNamespace(
        config_name=None, pooling="att", model="NAML", num_teacher_layers=12, num_student_layers=nl,
        bert_trainable_layer=list(cfg["trainable_layers"]), news_dim=D, news_query_vector_dim=200,
        user_query_vector_dim=200, num_teachers=T_, user_log_length=U, npratio=C - 1, num_words_title=L,
        user_log_mask=cfg["user_log_mask"], temperature=cfg["temperature"], coef=cfg["coef"], batch_size=B)


def test_model_module_is_a_drop_in():
    import model_bert
    z, P, cfg, inp = load_case("full_model_1.npz")
    T_ = len(inp[4])
    torch.cuda.set_device(0)
    model = model_bert.Model(_args(z, cfg, T_))
    sd = model.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == IFACE["state_dict"]
    assert sorted(k for k, p in model.named_parameters() if p.requires_grad) == sorted(IFACE["trainable_4layer_23"])
    assert hasattr(model, "student") and hasattr(model.student, "news_encoder") and len(model.teachers._modules) == T_
    model.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    out = model(t(inp[0]), t(inp[1]), t(inp[2]), t(inp[3]), [t(x) for x in inp[4]], [t(x) for x in inp[5]])
    total, distill, emb, target, score = out
    for got, key in ((total, "total"), (distill, "distill"), (emb, "emb"), (target, "target")):
        assert abs(got.item() - float(z[key])) < 1.6e-2 * max(1.0, abs(float(z[key]))), key
    assert score.shape == (inp[0].shape[0], inp[2].shape[1])
    opt = model_bert.TnrAdam(model, 1e-4)
    opt.zero_grad()
    total.backward()
    w = dict(model.named_parameters())
    k = "student.news_encoder.bert_model.bert.encoder.layer.3.output.dense.weight"
    assert w[k].grad is not None and abs(float(w[k].grad.norm()) - float(z["gnorm." + k])) < 6e-2 * float(z["gnorm." + k])
    assert w["student.news_encoder.bert_model.bert.embeddings.word_embeddings.weight"].grad is None
    assert w["teachers.0.pad_doc"].grad is None
    before = w[k].detach().clone()
    opt.step()
    assert not torch.equal(before, w[k].detach())
    # checkpoint round trip in the reference's dict layout (run.py:205-214)
    model.load_state_dict({k_: v.cpu() for k_, v in model.state_dict().items()})


def test_run_py_train_entry_point(tmp_path):
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tiny-newsrec_amd"))
    cmd = [sys.executable, "-u", os.path.join(ROOT, "tiny-newsrec_amd", "run.py"), "--mode", "train", "--synthetic", "True",
           "--enable_hvd", "False", "--batch_size", "8", "--epochs", "1", "--max_steps_per_epoch", "4", "--log_steps", "2",
           "--num_words_title", "30", "--news_dim", "256", "--num_student_layers", "2", "--bert_trainable_layer", "0", "1",
           "--num_teachers", "2", "--user_log_mask", "False", "--coef", "0.2", "--model", "NAML", "--model_type", "tnlrv3",
           "--model_dir", str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=os.path.join(ROOT, "tiny-newsrec_amd"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "train_loss" in r.stdout and "Model saved to" in r.stdout
    ck = torch.load(os.path.join(str(tmp_path), "epoch-1.pt"), map_location="cpu")
    assert set(ck) == {"model_state_dict", "category_dict", "word_dict", "subcategory_dict"}
    assert "student.user_encoder.pad_doc" in ck["model_state_dict"]


def test_bench_dp_path_under_torchrun_single_rank():
    """The driver launches bench.py with torch.distributed.run for N > 1; here N = 1 but with --force-dp, so the RCCL
    process group, the flat broadcast and the overlapped bucketed all-reduce + scaled AMSGrad all execute."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--force-dp", "--dp-sweep", "--no-configs"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and np.isfinite(d["config"]["final_loss"])
    assert d["larger_batch"]["batch_per_gpu"] == 128 and d["larger_batch"]["value"] > 0      # the informational 4 x batch leg
    # the data-parallel leg explains itself (what the first multi-GPU run needs to be read): defaults under data parallelism,
    # library version, the environment knobs RCCL reads, and the sweep over the knobs the build exposes
    dp = d["dp"]
    assert dp["backend"] == "nccl" and dp["wgrad_units_per_workgroup"] == 2 and dp["algo"] == "allreduce"
    assert dp["rccl_version"] and set(dp["env"]) >= {"NCCL_ALGO", "NCCL_PROTO"}
    assert len(dp["bucket_mb_in_completion_order"]) == 5 and len(dp["collectives_mb_in_launch_order"]) == 5
    sw = dp["sweep"]
    assert len(sw) == 8 and all("error" not in x and x["value"] > 0 for x in sw), sw
    assert {(x["wgrad_units"], x["buckets"], x["algo"]) for x in sw} == {(u, b, g) for u in (1, 2) for b in (5, 3) for g in ("allreduce", "rs_ag")}
    # the same step with the gradient exchange through the library's own C ABI (tnr_comm_*, round 6) instead of ProcessGroupNCCL
    cmd2 = [c for c in cmd if c != "--dp-sweep"] + ["--dp-algo", "native", "--no-larger-batch", "--no-other-dtype", "--dedup", "off"]
    cmd2[cmd2.index("29547")] = "29548"
    r2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r2.returncode == 0, r2.stdout[-1500:] + r2.stderr[-3000:]
    d2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["dp"]["algo"] == "native" and d2["value"] > 0 and d2["config"]["final_loss"] == d["config"]["final_loss"]


def test_bench_two_ranks_over_gloo_match_one_rank_on_the_concatenated_batch(tmp_path):
    """bench.py's OWN world > 1 path before the first 8-GPU lease runs it: `python bench.py --gpus 2` (self_launch ->
    torch.distributed.run -> two workers) with the test-only backend override - gloo, both ranks on the one GPU, B / 2 each -
    through the per-rank batches, the bucketed all-reduce with its per-bucket waits and scans (fp16 guard), max over ranks, the dp
    object, the warm-up autotune and --dp-sweep (rs_ag falls back to all-reduce off RCCL).  The parameters after the timed steps equal the one-rank run
    on the concatenated batches (horovod's average, Tiny-NewsRec/run.py:141-149) to the bound of tests/test_dp_gpu.py.  No
    throughput figure is taken from this."""
    steps, warm, lr = 3, 1, 1e-4
    outs = {}
    for tag, gpus, batch in (("two", 2, 8), ("one", 1, 16)):
        dump = str(tmp_path / (tag + ".npy"))
        env = dict(os.environ, TNR_BENCH_BACKEND="gloo", TNR_BENCH_SHARE_GPU="1", TNR_BENCH_DUMP_PARAMS=dump)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--batch", str(batch), "--steps", str(steps),
               "--warmup", str(warm), "--concat-batches", "--no-cpu-baseline", "--no-configs", "--no-larger-batch", "--no-other-dtype",
               "--dedup", "off", "--no-kernel-timing"] + (["--dp-sweep"] if gpus == 2 else [])
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == gpus and d["value"] > 0 and np.isfinite(d["config"]["final_loss"])
        assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp%d" % gpus
        outs[tag] = (d, np.load(dump))
    dp = outs["two"][0]["dp"]
    assert dp["backend"] == "gloo" and dp["algo"] == "allreduce"
    # the warm-up autotune timed {2, 1 wgrad units} x {5, 3 buckets}, then the best pair with the persistent GEMM grids sized for 248 /
    # 240 CUs (round 6), and the headline ran with the fastest (the same on both ranks)
    at = dp["autotune"]
    assert len(at["candidates"]) == 6 and all(c["ms_per_step"] > 0 for c in at["candidates"])
    assert sorted(c.get("cus", 0) for c in at["candidates"]) == [0, 0, 0, 0, 240, 248]
    assert at["chosen"] == min(at["candidates"], key=lambda c: c["ms_per_step"])
    assert (dp["wgrad_units_per_workgroup"], dp["buckets"]) == (at["chosen"]["wgrad_units"], at["chosen"]["buckets"])
    assert dp["cus_for_persistent_gemms"] == (at["chosen"].get("cus") or "all")
    assert len(dp["exposed_allreduce_ms_per_step_by_collective_rank0"]) == dp["buckets"] and dp["exposed_allreduce_ms_per_step_max_over_ranks"] > 0
    assert dp["fp16_tail_ms_per_step"]["amsgrad"] > 0 and dp["fp16_tail_ms_per_step"]["scans"] > 0
    sw = dp["sweep"]
    assert len(sw) == 8 and all("error" not in x and x["value"] > 0 and x["algo"] == "allreduce" for x in sw), sw
    assert "dp" not in outs["one"][0]
    p2, p1 = outs["two"][1].astype(np.float64), outs["one"][1].astype(np.float64)
    diff = np.abs(p2 - p1)
    n_upd = steps + warm
    print("bench, 2 ranks (gloo) vs 1 rank: |param diff| mean %.2e max %.2e after %d steps at lr %g" % (diff.mean(), diff.max(), n_upd, lr))
    assert diff.mean() < 0.02 * lr and diff.max() <= 2.05 * lr * n_upd


def test_construction_time_init_and_pretrained_import(tmp_path):
    """Model(args) starts from the reference's construction-time distributions, and --model_name pointing at a
    unilm2-layout checkpoint fills the student's encoder like from_pretrained does (model_bert.py:109-114,
    tnlrv3/convert_state_dict.py, tnlrv3/modeling.py:90-118): first num_student_layers layers, fused qkv split,
    position table fitted.  The imported encoder is then checked end to end against the oracle."""
    import model_bert
    from helpers import unilm_checkpoint
    from oracle import newsrec_oracle as O
    from tnlrv3 import convert_state_dict as C
    H, A, I, vocab, L, D = 768, 12, 3072, 600, 30, 64
    cfgp = tmp_path / "config.json"
    cfgp.write_text(json.dumps(dict(hidden_size=H, num_attention_heads=A, intermediate_size=I, vocab_size=vocab,
                                    max_position_embeddings=48, type_vocab_size=2, layer_norm_eps=1e-12)))
    raw = unilm_checkpoint(5, H, 3, A, I, vocab, 32)          # 3 layers and 32 positions in the checkpoint
    ck = tmp_path / "unilm2-tiny.bin"
    torch.save(raw, ck)
    args = types.SimpleNamespace(
        config_name=str(cfgp), model_name=str(ck), pooling="att", model="NAML", num_teacher_layers=12, num_student_layers=2,
        bert_trainable_layer=[1], news_dim=D, news_query_vector_dim=200, user_query_vector_dim=200, num_teachers=1,
        user_log_length=4, npratio=1, num_words_title=L, user_log_mask=False, temperature=1.0, coef=0.2, batch_size=2)
    torch.cuda.set_device(0)
    model = model_bert.Model(args)
    missing, unexpected = model.pretrained_report
    assert sorted(missing) == ["student.news_encoder.bert_model.classifier.bias", "student.news_encoder.bert_model.classifier.weight"]
    assert "cls.predictions.bias" in unexpected and any(k.startswith("bert.encoder.layer.2.") for k in unexpected)
    sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    pfx = "student.news_encoder.bert_model."
    conv = C.load_model(dict(raw))
    for k in ("bert.encoder.layer.1.attention.self.key.weight", "bert.encoder.layer.0.attention.self.value.bias",
              "bert.rel_pos_bias.weight", "bert.embeddings.word_embeddings.weight", "bert.pooler.dense.weight"):
        assert np.array_equal(sd[pfx + k], conv[k].numpy()), k
    assert not sd[pfx + "bert.encoder.layer.0.attention.self.key.bias"].any()
    pos = sd[pfx + "bert.embeddings.position_embeddings.weight"]
    assert pos.shape == (48, H) and np.array_equal(pos[:32], raw["bert.embeddings.position_embeddings.weight"].numpy())
    assert 0.015 < pos[32:].std() < 0.025                                  # new rows ~ N(0, 0.02)
    # heads: nn.Linear default / pad_doc U(-1,1) / Xavier transform (none is left at zero)
    w = sd["student.news_encoder.dense.weight"]
    assert abs(w).max() <= 1 / np.sqrt(H) + 1e-6 and w.std() > 0.5 / np.sqrt(3 * H)
    assert 0.4 < sd["student.user_encoder.pad_doc"].std() < 0.75
    b = np.sqrt(6.0 / (2 * D))
    assert abs(sd["transform_matrix.0.weight"]).max() <= b + 1e-6 and not sd["transform_matrix.0.bias"].any()
    # encoder forward with the imported weights == oracle on the same weights
    ids = np.zeros((6, 2 * L), dtype=np.int64)
    rng = np.random.RandomState(0)
    for r in range(6):
        n = 5 + 4 * r
        ids[r, :n] = rng.randint(1, vocab, n)
        ids[r, L:L + n] = 1
    got = model.engine.encode(torch.from_numpy(ids).cuda(), 6).cpu().numpy()
    ref, _ = O.news_encoder_fwd(sd, ids, 2, A, None)
    print("imported-encoder max|err| %.3e (|ref| max %.3e)" % (np.abs(got - ref).max(), np.abs(ref).max()))
    assert np.abs(got - ref).max() <= 3e-2 * np.abs(ref).max() + 1e-5


def test_plmnr_modelbert_surface():
    """PLM-NR/model_bert.py:178-207 + run.py:104-106 on the engine: ModelBert(args)(history, mask, candidate, label) ->
    (loss, score), keys without the 'student.' prefix, two learning rates."""
    import model_bert
    from helpers import load_plmnr_case
    z, P, cfg, inp = load_plmnr_case()
    seed, B, _, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
    args = types.SimpleNamespace(
        config_name=None, model_name=None, pooling="att", model="NAML", num_hidden_layers=nl, num_teacher_layers=12,
        num_student_layers=4, bert_trainable_layer=[0, 1], news_dim=D, news_query_vector_dim=200, user_query_vector_dim=200,
        num_teachers=4, user_log_length=U, npratio=C - 1, num_words_title=L, user_log_mask=False, temperature=1.0, coef=0.2,
        batch_size=B, num_attention_heads=16)
    torch.cuda.set_device(0)
    model = model_bert.ModelBert(args)
    sd = model.state_dict()
    assert all(not k.startswith("student.") and not k.startswith("teachers.") for k in sd)
    assert "news_encoder.bert_model.bert.encoder.layer.1.output.dense.weight" in sd and "user_encoder.pad_doc" in sd
    model.load_state_dict({k[len("student."):]: torch.from_numpy(v) for k, v in P.items()})
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    opt = model_bert.TnrAdam(model, float(z["lrs"][1]), pretrain_lr=float(z["lrs"][0]))
    for step in range(2):
        loss, score = model(t(inp[0]), t(inp[1]), t(inp[2]), t(inp[3]))
        assert abs(loss.item() - float(z["loss%d" % step])) < 1.6e-2 * max(1.0, float(z["loss%d" % step]))
        opt.zero_grad()
        loss.backward()
        opt.step()
    w = dict(model.named_parameters())
    assert w["news_encoder.dense.weight"].grad is not None and w["news_encoder.bert_model.bert.embeddings.word_embeddings.weight"].grad is None


@pytest.mark.parametrize("model_type", ["bert", "roberta"])
def test_plmnr_modelbert_bert_and_roberta_model_types(model_type, tmp_path):
    """PLM-NR --model_type bert / roberta (PLM-NR/utils.py:17-21): ModelBert with transformers BertModel / RobertaModel as the
    encoder -- the module tree and state_dict keys of the reference (parameters directly under bert_model.*, no rel-pos bias, no
    classification head), loss / scores of the reference's own run (fp16 bound), the freeze pattern of PLM-NR/run.py:119-124, and
    the pretrained-checkpoint import of BertModel.from_pretrained (prefixed keys, old LayerNorm names)."""
    import model_bert
    from helpers import load_plmnr_hf_case
    z, P, cfg, inp = load_plmnr_hf_case(model_type)
    seed, B, _, U, C, L, D, A, nl = [int(x) for x in z["meta"]]
    cj = dict(hidden_size=768, num_attention_heads=A, intermediate_size=3072, vocab_size=cfg["vocab"], max_position_embeddings=cfg["max_pos"],
              type_vocab_size=cfg["type_vocab"], layer_norm_eps=cfg["ln_eps"], pad_token_id=1 if model_type == "roberta" else 0)
    (tmp_path / "config.json").write_text(json.dumps(cj))
    args = types.SimpleNamespace(
        config_name=str(tmp_path / "config.json"), model_name=None, model_type=model_type, pooling="att", model="NAML", num_hidden_layers=nl,
        num_teacher_layers=12, num_student_layers=4, bert_trainable_layer=[0, 1], news_dim=D, news_query_vector_dim=200,
        user_query_vector_dim=200, num_teachers=4, user_log_length=U, npratio=C - 1, num_words_title=L, user_log_mask=False,
        temperature=1.0, coef=0.2, batch_size=B, num_attention_heads=16)
    torch.cuda.set_device(0)
    model = model_bert.ModelBert(args)
    sd = model.state_dict()
    assert sorted(sd) == sorted(str(k) for k in z["keys"])                 # the reference's keys, nothing else
    assert float(model.engine.rel.abs().max()) == 0.0                       # no relative-position bias in these encoders
    ref_sd = {k[len("student."):].replace(".bert_model.bert.", ".bert_model."): torch.from_numpy(v) for k, v in P.items()
              if not (k.endswith("rel_pos_bias.weight") or ".bert_model.classifier." in k)}
    ref_sd["news_encoder.bert_model.embeddings.position_ids"] = torch.arange(cfg["max_pos"]).unsqueeze(0)     # transformers 3.0.2 buffer
    model.load_state_dict(ref_sd)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    loss, score = model(t(inp[0]), t(inp[1]), t(inp[2]), t(inp[3]))
    serr = np.abs(score.detach().cpu().numpy() - z["score0"]).max()
    print("\n[plmnr %s] loss %.6f ref %.6f ; score max|err| %.2e (|ref| max %.2f)" % (model_type, loss.item(), float(z["loss0"]), serr, np.abs(z["score0"]).max()))
    assert abs(loss.item() - float(z["loss0"])) <= 1e-3 * max(1.0, float(z["loss0"]))
    assert serr <= 1e-3 * max(1.0, np.abs(z["score0"]).max())
    loss.backward()
    w = dict(model.named_parameters())
    assert w["news_encoder.bert_model.embeddings.word_embeddings.weight"].grad is None          # frozen, PLM-NR/run.py:119-121
    for n in [str(x) for x in z["grad_names"]]:
        if n.endswith("self.key.bias") or n.endswith("att_fc2.bias"):
            continue
        g, gn = w[n].grad.cpu().numpy(), float(z["gnorm." + n])
        assert abs(np.sqrt((g.astype(np.float64) ** 2).sum()) - gn) <= 1.5e-2 * gn + 1e-7, n
        ref = z["gval." + n]
        assert np.abs(g.reshape(-1)[z["gidx." + n]] - ref).max() <= 2e-2 * np.abs(ref).max() + 1e-7, n
    # BertModel.from_pretrained: a checkpoint with the "bert." / "roberta." prefix and pre-1.0 LayerNorm names
    ck = {}
    for k, v in ref_sd.items():
        if ".bert_model." in k and "position_ids" not in k:
            kk = model_type + "." + k.split(".bert_model.")[1]
            ck[kk.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")] = v + 1.0
    ck["cls.predictions.bias"] = torch.zeros(4)
    torch.save(ck, tmp_path / "pytorch_model.bin")
    missing, unexpected = model.load_pretrained(str(tmp_path / "pytorch_model.bin"))
    assert missing == [] and unexpected == ["cls.predictions.bias"]
    k0 = "news_encoder.bert_model.encoder.layer.1.output.LayerNorm.weight"
    assert torch.equal(model.state_dict()[k0].cpu(), ref_sd[k0] + 1.0)


def test_run_py_train_from_mind_format_files(tmp_path):
    """`python run.py --mode train` on real-format inputs (news.tsv through the BERT wordpiece tokenizer, behaviors_np4_*.tsv
    shards through the TF-free streamer, teacher-embedding pickles, PLM-NR teacher checkpoints): the demo.sh train flow
    end to end, resident tables + in-batch de-duplication on, one epoch, checkpoint written in the reference's layout."""
    import pickle
    import hashinit
    from helpers import FULL, state_shapes
    data = os.path.join(GOLDEN, "data")
    words = sorted({w for ln in open(os.path.join(data, "news.tsv")) for w in ln.split("\t")[3].lower().split()})
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    (tmp_path / "config.json").write_text(json.dumps(dict(
        hidden_size=768, num_attention_heads=12, intermediate_size=3072, vocab_size=len(vocab), max_position_embeddings=64,
        type_vocab_size=2, layer_norm_eps=1e-12)))
    n_news = sum(1 for _ in open(os.path.join(data, "news.tsv")))
    embs, ckpts = [], []
    for i in range(2):
        p = tmp_path / ("teacher_emb_%d.pkl" % i)
        with open(p, "wb") as f:
            pickle.dump(hashinit.hash_normal(50 + i, "temb", (n_news + 1, 256)), f)
        embs.append(str(p))
        sd = hashinit.init_state_dict(60 + i, {k[len("student."):]: v for k, v in state_shapes(FULL, 1, 256, 0).items()
                                               if k.startswith("student.user_encoder.")})
        ck = tmp_path / ("teacher_%d.pt" % i)
        torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}}, ck)
        ckpts.append(str(ck))
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tiny-newsrec_amd"))
    cmd = [sys.executable, "-u", os.path.join(ROOT, "tiny-newsrec_amd", "run.py"), "--mode", "train", "--enable_hvd", "False",
           "--train_data_dir", data, "--filename_pat", "behaviors_np4_*.tsv", "--batch_size", "4", "--epochs", "1",
           "--log_steps", "1", "--num_words_title", "30", "--news_dim", "256", "--num_student_layers", "2",
           "--bert_trainable_layer", "0", "1", "--num_teachers", "2", "--user_log_mask", "False", "--coef", "0.2",
           "--model", "NAML", "--model_type", "tnlrv3", "--model_dir", str(tmp_path / "out"), "--tokenizer_name",
           str(tmp_path / "vocab.txt"), "--config_name", str(tmp_path / "config.json"), "--model_name", str(tmp_path / "none.bin"), "--allow_random_init", "True",
           "--teacher_emb_paths"] + embs + ["--teacher_ckpts"] + ckpts
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=os.path.join(ROOT, "tiny-newsrec_amd"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    log = r.stdout + r.stderr
    assert "train_loss" in log and "nan" not in log.lower()
    ck = torch.load(str(tmp_path / "out" / "epoch-1.pt"), map_location="cpu")
    assert set(ck) >= {"model_state_dict", "category_dict", "subcategory_dict", "word_dict"}
    sd = ck["model_state_dict"]
    assert {k: list(v.shape) for k, v in sd.items() if "word_embeddings" not in k and "position_embeddings" not in k} == \
        {k: v for k, v in {**IFACE_2L(), }.items() if "word_embeddings" not in k and "position_embeddings" not in k}
    # the teachers' user encoders came from the PLM-NR checkpoints (run.py:61-70) and stayed frozen
    t0 = torch.load(ckpts[0], map_location="cpu")["model_state_dict"]
    assert torch.equal(sd["teachers.0.attn.att_fc1.weight"], t0["user_encoder.attn.att_fc1.weight"])


def IFACE_2L():
    """state_dict schema of a 2-layer student + 2 teachers, derived from the captured 4-layer / 4-teacher one."""
    out = {}
    for k, v in IFACE["state_dict"].items():
        if ".encoder.layer.2." in k or ".encoder.layer.3." in k or k.startswith("teachers.2.") or k.startswith("teachers.3.") \
                or k.startswith("transform_matrix.2.") or k.startswith("transform_matrix.3."):
            continue
        out[k] = v
    return out


def test_run_py_plmnr_train_and_teacher_pipeline(tmp_path):
    """`run.py --mode train --num_teachers 0` = PLM-NR/run.py's train(): CE fine-tuning of ModelBert with PLM-NR checkpoint
    keys, optionally initialised from a first-stage student (two learning rates); its checkpoint then serves as a teacher
    checkpoint of the Tiny-NewsRec flow (run.py:61-70 reads user_encoder.* from it)."""
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tiny-newsrec_amd"))
    tok_args = _tok_cfg_args(tmp_path)
    first = {"student." + k: v for k, v in _plmnr_sd(2).items() if k.startswith("news_encoder.")}
    torch.save({"model_state_dict": first}, str(tmp_path / "first_stage_2_layer.pt"))
    cmd = [sys.executable, "-u", os.path.join(ROOT, "tiny-newsrec_amd", "run.py"), "--mode", "train", "--synthetic", "False",
           "--enable_hvd", "False", "--num_teachers", "0", "--num_hidden_layers", "2", "--bert_trainable_layer", "0", "1",
           "--batch_size", "4", "--epochs", "1", "--log_steps", "1", "--use_pretrain_model", "True", "--pretrain_model_path",
           str(tmp_path / "first_stage_2_layer.pt"), "--pretrain_lr", "1e-6", "--lr", "1e-4", "--model_dir", str(tmp_path / "out"),
           "--train_data_dir", os.path.join(GOLDEN, "data"), "--filename_pat", "behaviors_np4_*.tsv"] + tok_args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=os.path.join(ROOT, "tiny-newsrec_amd"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    sd = torch.load(str(tmp_path / "out" / "epoch-1.pt"), map_location="cpu")["model_state_dict"]
    assert "user_encoder.attn.att_fc1.weight" in sd and "news_encoder.dense.weight" in sd and not any(k.startswith("student.") for k in sd)
    # the pretrained encoder moved by ~pretrain_lr per step, the fresh user encoder by ~lr per step (3 steps)
    w0 = first["student.news_encoder.bert_model.bert.encoder.layer.1.output.dense.weight"]
    dw = (sd["news_encoder.bert_model.bert.encoder.layer.1.output.dense.weight"] - w0).abs()
    assert 0 < dw.max() < 1e-5
    dd = (sd["news_encoder.dense.weight"] - first["student.news_encoder.dense.weight"]).abs()
    assert 0 < dd.max() < 1e-5                               # pooling / dense of the news encoder were pretrained too


def _plmnr_sd(nl):
    import hashinit
    from helpers import FULL, state_shapes
    shapes = {k[len("student."):]: v for k, v in state_shapes(dict(FULL, vocab=_VOCAB_N), nl, 256, 0).items() if k.startswith("student.")}
    shapes = {k: ((64, 768) if k.endswith("position_embeddings.weight") else v) for k, v in shapes.items()}
    return {k: torch.from_numpy(v) for k, v in hashinit.init_state_dict(9, shapes).items()}


_VOCAB_N = 0


def _tok_cfg_args(tmp_path):
    global _VOCAB_N
    data = os.path.join(GOLDEN, "data")
    words = sorted({w for ln in open(os.path.join(data, "news.tsv")) for w in ln.split("\t")[3].lower().split()})
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    (tmp_path / "config.json").write_text(json.dumps(dict(
        hidden_size=768, num_attention_heads=12, intermediate_size=3072, vocab_size=len(vocab), max_position_embeddings=64,
        type_vocab_size=2, layer_norm_eps=1e-12)))
    _VOCAB_N = len(vocab)
    return ["--tokenizer_name", str(tmp_path / "vocab.txt"), "--config_name", str(tmp_path / "config.json"), "--model_name",
            str(tmp_path / "none.bin"), "--allow_random_init", "True", "--num_words_title", "30", "--news_dim", "256", "--user_log_mask", "False",
            "--model", "NAML", "--model_type", "tnlrv3"]


def test_reference_feed_from_the_producer_thread_equals_the_host_tensors():
    """DataLoaderTrain(resident=False) = the reference's 6-tuple (dataloader.py:151-172).  From the producer thread its tensors
    leave as one pinned staging buffer + one asynchronous copy on the producer's stream (round 6) instead of 4 + 2 T pageable
    `.cuda()` copies: every batch of an epoch - dtypes, shapes, values, the short last batch - equals the CPU loader's on the same
    files and the same label draws."""
    import random
    import types as _types
    import dataloader
    import hashinit
    z = np.load(os.path.join(GOLDEN, "datapath.npz"))
    news_index = {str(k): int(v) for k, v in zip(z["news_ids"], z["news_index"])}
    comb = z["news_combined"]
    temb = [hashinit.hash_normal(11, "dp_temb%d" % i, (comb.shape[0], 8)) for i in range(2)]
    args = _types.SimpleNamespace(npratio=4, user_log_length=50, batch_size=5, shuffle_buffer_size=7, num_teachers=2)
    mk = lambda gpu: dataloader.DataLoaderTrain(os.path.join(GOLDEN, "data"), "behaviors_np4_*.tsv", args, 1, 0, 0, news_index, comb, temb,
                                                enable_prefetch=True, enable_shuffle=False, enable_gpu=gpu, resident=False)
    random.seed(5)
    host = [b for b in mk(False)]
    random.seed(5)
    dev = [b for b in mk(True)]
    assert len(host) == len(dev) == 3 and isinstance(dev[0], dataloader.RowBatch)
    torch.cuda.synchronize()
    for hb, db in zip(host, dev):
        flat_h = list(hb[:4]) + list(hb[4]) + list(hb[5])
        flat_d = list(db[:4]) + list(db[4]) + list(db[5])
        assert len(flat_h) == len(flat_d) == 8
        for a, b in zip(flat_h, flat_d):
            assert b.is_cuda and a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b.cpu())


@pytest.mark.parametrize("teachers", [0, 2])
def test_run_py_reference_feed_mode_without_resident_tables(tmp_path, teachers):
    """--resident_tables False = the reference's own feed (dataloader.py:151-172: gathered int64 token rows and fp32 teacher
    rows cross PCIe every step) through Model.forward / ModelBert.forward instead of forward_indexed; both objectives.
    Also: a --model_name that does not exist must fail loudly unless --allow_random_init / --synthetic is given."""
    import pickle
    from helpers import FULL, state_shapes
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tiny-newsrec_amd"))
    tok_args = _tok_cfg_args(tmp_path)
    data = os.path.join(GOLDEN, "data")
    n_news = sum(1 for _ in open(os.path.join(data, "news.tsv")))
    extra = []
    if teachers:
        import hashinit
        embs, ckpts = [], []
        for i in range(teachers):
            p = tmp_path / ("temb_%d.pkl" % i)
            with open(p, "wb") as f:
                pickle.dump(hashinit.hash_normal(40 + i, "emb", (n_news + 1, 256), 1.0), f)
            embs.append(str(p))
            sd = hashinit.init_state_dict(60 + i, {k[len("student."):]: v for k, v in state_shapes(FULL, 1, 256, 0).items()
                                                   if k.startswith("student.user_encoder.")})
            ck = tmp_path / ("teacher_%d.pt" % i)
            torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}}, ck)
            ckpts.append(str(ck))
        extra = ["--num_student_layers", "2", "--coef", "0.2", "--teacher_emb_paths"] + embs + ["--teacher_ckpts"] + ckpts
    else:
        extra = ["--num_hidden_layers", "2"]
    cmd = [sys.executable, "-u", os.path.join(ROOT, "tiny-newsrec_amd", "run.py"), "--mode", "train", "--enable_hvd", "False",
           "--resident_tables", "False", "--num_teachers", str(teachers), "--bert_trainable_layer", "0", "1", "--batch_size", "4",
           "--epochs", "1", "--log_steps", "1", "--model_dir", str(tmp_path / "out"), "--train_data_dir", data,
           "--filename_pat", "behaviors_np4_*.tsv"] + tok_args + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=os.path.join(ROOT, "tiny-newsrec_amd"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    log = r.stdout + r.stderr
    assert "train_loss" in log and "nan" not in log.lower() and os.path.exists(str(tmp_path / "out" / "epoch-1.pt"))
    if teachers == 0:
        i = cmd.index("--allow_random_init")
        strict = cmd[:i] + cmd[i + 2:]
        r2 = subprocess.run(strict, env=env, capture_output=True, text=True, timeout=900, cwd=os.path.join(ROOT, "tiny-newsrec_amd"))
        assert r2.returncode != 0 and "does not exist" in (r2.stdout + r2.stderr)
