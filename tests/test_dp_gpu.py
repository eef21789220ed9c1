"""Data-parallel equivalence of the ENGINE (level 3; reference: Tiny-NewsRec/run.py:141-149 --
hvd.DistributedOptimizer(op=Average)): two ranks with B/2 impressions each, gradients summed bucket by bucket from the
backward's hook and scaled by 1/W, must give the gradient of the concatenated B-impression batch, and every rank must hold
identical parameters after step().  The ranks run as two processes sharing cuda:0 (gloo transport; on a multi-GPU node
bench.py uses RCCL -- the bucket logic, launch points and scaling are the same code)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "dp_rank.py")


def _launch(tmp_path, world, dtype, B, steps, tag, port):
    out = str(tmp_path / (tag + "_rank%d.npz"))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, WORKER, out, dtype, str(B), str(steps)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=900)[0] for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    return [np.load(out % r) for r in range(world)]


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_two_ranks_equal_one_rank_with_the_concatenated_batch(tmp_path, dtype):
    B, steps, lr = 4, 2, 1e-4
    two = _launch(tmp_path, 2, dtype, B, steps, "w2", 29611)
    one = _launch(tmp_path, 1, dtype, B, steps, "w1", 29612)[0]
    # (1) both ranks hold the same summed gradient and bit-identical parameters after every step
    assert np.array_equal(two[0]["grads"], two[1]["grads"])
    assert np.array_equal(two[0]["params"], two[1]["params"])
    # (2) first step: mean of the two half-batch gradients == gradient of the whole batch (same weights on both sides)
    g2, g1 = two[0]["grads"][0].astype(np.float64), one["grads"][0].astype(np.float64)
    cut = int(one["head0"])
    for name, sl in (("encoder + pooling", slice(0, cut)), ("heads", slice(cut, None))):
        rel = np.linalg.norm(g2[sl] - g1[sl]) / np.linalg.norm(g1[sl])
        print("%s %s: rel. L2 difference of the averaged gradient %.2e" % (dtype, name, rel))
        assert rel < (1e-3 if dtype == "fp16" else 8e-3), (name, rel)
    # (3) parameters after two AMSGrad steps: Adam normalises every update to ~lr, so tiny gradient differences can move
    # an element by a fraction of lr at most where the gradient is not ~0; bound the mean and the worst element
    dp = np.abs(two[0]["params"].astype(np.float64) - one["params"].astype(np.float64))
    print("%s: |param diff| mean %.2e max %.2e (lr %.0e)" % (dtype, dp.mean(), dp.max(), lr))
    assert dp.mean() < 0.02 * lr and dp.max() <= 2.05 * lr * steps
