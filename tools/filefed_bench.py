"""File-fed end-to-end throughput of `run.py --mode train` on one GPU (round-3 review, item 7): the benchmark hands the engine
pre-drawn resident indices; here the WHOLE reference flow runs - MIND-format news.tsv through the wordpiece tokenizer, a raw
behaviors.tsv through split_file.py into behaviors_np4_*.tsv, the TF-free streamer + DataLoaderTrain's producer thread (TSV
decode, label draw, index / row gathers, H2D), teacher-embedding pickles and teacher checkpoints - and the log's impressions/s
is compared with bench.py's figure for the matching mode.
    python tools/filefed_bench.py [--lines 200000] [--news 51282] [--steps 1500]
Modes: "default" = run.py's defaults (resident tables + in-batch de-duplication + frozen-layer cache), "plain" = all three off
(the reference's feed: gathered int64 token rows and fp32 teacher rows shipped every step), "resident" = resident tables only
(bench.py's headline feed).  Prints one JSON line."""
import argparse, json, os, pickle, re, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tiny-newsrec_amd")
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_corpus(d, n_news, n_lines, seed=5):
    rs = np.random.RandomState(seed)
    words = ["w%04d" % i for i in range(5000)]
    with open(os.path.join(d, "vocab.txt"), "w") as f:
        f.write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words) + "\n")
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, vocab_size=5005, max_position_embeddings=64,
                       type_vocab_size=2, layer_norm_eps=1e-12), f)
    lens = np.clip(np.rint(rs.normal(12, 4, n_news)), 1, 28).astype(int)          # + [CLS] / [SEP] -> mean 14 of 30
    with open(os.path.join(d, "news.tsv"), "w") as f:
        for i in range(n_news):
            t = " ".join(words[j] for j in rs.randint(0, 5000, lens[i]))
            f.write("N%d\tcat%d\tsub%d\t%s\tabs\turl\t[]\t[]\n" % (i + 1, i % 17, i % 250, t))
    # raw MIND behaviors with EXACTLY the statistics of bench.py's synthetic impressions (synth.impressions: history lengths,
    # popularity law, uniform candidates), so that the two figures compare like with like - the share of distinct news per batch
    # decides what the de-duplicated modes cost; one clicked + four skipped candidates per impression = one training line each
    import synth
    hidx, hmask, cidx, label = synth.impressions(1235, n_lines, n_news, 50, 5)
    with open(os.path.join(d, "behaviors.tsv"), "w") as f:
        for i in range(n_lines):
            hist = " ".join("N%d" % x for x in hidx[i][hmask[i] > 0])
            imp = " ".join("N%d-%d" % (c, 1 if j == label[i] else 0) for j, c in enumerate(cidx[i]))
            f.write("%d\tU%d\t11/15/2019 8:55:22 AM\t%s\t%s\n" % (i, i % 50000, hist, imp))
    import split_file
    paths, n = split_file.split(os.path.join(d, "behaviors.tsv"), 1, 4, seed=7)
    import hashinit
    from helpers import FULL, state_shapes
    import torch
    embs, ckpts = [], []
    for i in range(4):
        p = os.path.join(d, "teacher_emb_%d.pkl" % i)
        with open(p, "wb") as f:
            pickle.dump(rs.standard_normal((n_news + 1, 256)).astype(np.float32), f)
        embs.append(p)
        sd = hashinit.init_state_dict(60 + i, {k_[len("student."):]: v for k_, v in state_shapes(FULL, 1, 256, 0).items()
                                               if k_.startswith("student.user_encoder.")})
        ck = os.path.join(d, "teacher_%d.pt" % i)
        torch.save({"model_state_dict": {k_: torch.from_numpy(v) for k_, v in sd.items()}}, ck)
        ckpts.append(ck)
    return n, embs, ckpts


def run_mode(d, embs, ckpts, steps, extra, log_steps=100):
    cmd = [sys.executable, "-u", os.path.join(PKG, "run.py"), "--mode", "train", "--enable_hvd", "False", "--train_data_dir", d,
           "--filename_pat", "behaviors_np4_*.tsv", "--batch_size", "32", "--epochs", "1", "--log_steps", str(log_steps),
           "--max_steps_per_epoch", str(steps), "--num_words_title", "30", "--news_dim", "256", "--num_student_layers", "4",
           "--bert_trainable_layer", "2", "3", "--num_teachers", "4", "--user_log_mask", "False", "--coef", "0.2", "--model", "NAML",
           "--model_type", "tnlrv3", "--model_dir", os.path.join(d, "out"), "--tokenizer_name", os.path.join(d, "vocab.txt"),
           "--config_name", os.path.join(d, "config.json"), "--model_name", os.path.join(d, "none.bin"), "--allow_random_init", "True",
           "--teacher_emb_paths"] + embs + ["--teacher_ckpts"] + ckpts + extra
    t0 = time.time()
    r = subprocess.run(cmd, env=dict(os.environ, PYTHONPATH=PKG), capture_output=True, text=True, cwd=PKG)
    log = r.stdout + r.stderr
    if r.returncode != 0:
        return {"error": log[-1500:]}
    pts = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"Ed: (\d+), train_loss: [-\d.naife]+, acc: [\d.naife]+, ([\d.]+) impressions/s", log)]
    pts = [(ed, ed / v) for ed, v in pts if ed > 0 and v > 0]            # (impressions done, seconds since the epoch started)
    out = {"wall_s": round(time.time() - t0, 1), "log_points": len(pts)}
    if len(pts) >= 3:
        (e0, s0), (e1, s1) = pts[len(pts) // 4], pts[-1]                 # steady state: the last three quarters of the epoch
        out["impressions_per_s"] = round((e1 - e0) / max(s1 - s0, 1e-9), 1)
        out["impressions_per_s_whole_epoch"] = round(e1 / s1, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=200000)
    ap.add_argument("--news", type=int, default=51282)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--keep", default=None, help="directory to build the corpus in (kept); default: a temporary one")
    a = ap.parse_args()
    d = a.keep or tempfile.mkdtemp(prefix="tnr_filefed_")
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    n, embs, ckpts = make_corpus(d, a.news, a.lines)
    res = {"news": a.news, "train_lines": n, "steps_per_mode": a.steps, "corpus_build_s": round(time.time() - t0, 1), "modes": {}}
    off = ["--resident_tables", "False", "--cache_frozen_layers", "False", "--dedup_news", "False"]
    for name, extra in (("default (resident + dedup + frozen-layer cache)", []),
                        ("resident tables only (bench.py's headline feed)", ["--cache_frozen_layers", "False", "--dedup_news", "False"]),
                        ("plain (the reference's feed: gathered rows shipped every step)", off)):
        res["modes"][name] = run_mode(d, embs, ckpts, a.steps, extra)
        print(name, res["modes"][name], file=sys.stderr, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
