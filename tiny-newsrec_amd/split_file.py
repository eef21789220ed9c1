"""MIND behaviors.tsv -> behaviors_np4_{i}.tsv, the train-sample format the loader reads (reference
split_file.py:6-43, TF-free and parameterised): one line per positive click with `npratio` negatives sampled
from the same impression (with replacement only when there are too few), all lines shuffled, then dealt
round-robin into `n_files` shards (one per data-parallel worker, streaming.get_worker_files takes files[rank::world]).

    python split_file.py --behaviors ../MIND/MINDlarge_train/behaviors.tsv --n_files 8

With the same `random` seed and inputs the output is identical to the reference's (same draw order)."""
import argparse
import os
import random


def get_sample(all_element, num_sample):
    if num_sample > len(all_element):
        return random.sample(all_element * (num_sample // len(all_element) + 1), num_sample)
    return random.sample(all_element, num_sample)


def make_samples(lines, npratio=4):
    out = []
    for line in lines:
        iid, uid, time, history, imp = line.strip("\n").split("\t")
        pos, neg = [], []
        for item in imp.split(" "):
            nid, label = item.split("-")
            (pos if int(label) == 1 else neg).append(nid) if int(label) in (0, 1) else None
        if not pos:
            continue
        for pos_id in pos:
            out.append("\t".join([iid, uid, time, history, pos_id, " ".join(get_sample(neg, npratio))]) + "\n")
    return out


def split(behaviors_path, n_files, npratio=4, out_dir=None, seed=None):
    if seed is not None:
        random.seed(seed)
    with open(behaviors_path) as f:
        samples = make_samples(f, npratio)
    random.shuffle(samples)
    out_dir = out_dir or os.path.dirname(behaviors_path)
    paths = []
    for i in range(n_files):
        p = os.path.join(out_dir, "behaviors_np%d_%d.tsv" % (npratio, i))
        with open(p, "w") as f:
            f.writelines(samples[i::n_files])
        paths.append(p)
    return paths, len(samples)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--behaviors", default="./MIND/MINDlarge_train/behaviors.tsv")
    ap.add_argument("--n_files", type=int, default=4)
    ap.add_argument("--npratio", type=int, default=4)
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    paths, n = split(a.behaviors, a.n_files, a.npratio, seed=a.seed)
    print(n)
