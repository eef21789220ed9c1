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


def _launch(tmp_path, world, dtype, B, steps, tag, port, mode="stage2"):
    out = str(tmp_path / (tag + "_rank%d.npz"))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, WORKER, out, dtype, str(B), str(steps), mode], env=env,
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


def test_stage1_two_ranks_equal_one_rank_with_the_concatenated_batch(tmp_path):
    """post_train_kd.py's data-parallel path (stage1.py:143-154: the title pass writes the gradients, the body pass accumulates
    and fires the buckets, so title + body gradients are summed BEFORE the all-reduce): two ranks x B/2 == one rank x B."""
    B, steps = 4, 2
    two = _launch(tmp_path, 2, "fp16", B, steps, "s1w2", 29621, "stage1")
    one = _launch(tmp_path, 1, "fp16", B, steps, "s1w1", 29622, "stage1")[0]
    assert np.array_equal(two[0]["grads"], two[1]["grads"]) and np.array_equal(two[0]["params"], two[1]["params"])
    g2, g1 = two[0]["grads"][0].astype(np.float64), one["grads"][0].astype(np.float64)
    cut = int(one["head0"])
    for name, sl in (("encoder + pooling", slice(0, cut)), ("heads", slice(cut, None))):
        rel = np.linalg.norm(g2[sl] - g1[sl]) / np.linalg.norm(g1[sl])
        print("stage 1 %s: rel. L2 difference of the averaged gradient %.2e" % (name, rel))
        assert rel < 1e-3, (name, rel)
    dp = np.abs(two[0]["params"].astype(np.float64) - one["params"].astype(np.float64))
    print("stage 1: |param diff| mean %.2e max %.2e" % (dp.mean(), dp.max()))
    assert dp.mean() < 0.02 * 1e-5 and dp.max() <= 2.05 * 1e-5 * steps


class _FakeSync:
    """A GradSync whose collectives are already complete: Engine.step(sync=...) must take the bucket-by-bucket path."""

    def __init__(self, ranges):
        self.ranges, self.pending, self.waited, self.scale = list(ranges), {b: None for b in range(len(ranges))}, [], 0.5

    def wait_bucket(self, b):
        self.pending.pop(b, None)
        self.waited.append(b)

    def wait(self):
        for b in list(self.pending):
            self.wait_bucket(b)


@pytest.mark.parametrize("rates", [(1e-4, None, None), (1e-5, 1e-6, None), (1e-4, 2e-5, 3e-5)])
def test_bucketwise_optimiser_step_is_bit_identical_to_one_launch(rates):
    """Engine.step(sync=...) runs AMSGrad bucket by bucket in completion order (bucket ranges intersected with the learning-rate
    ranges, then a sweep over the alignment gaps): every element must get exactly one update with ITS rate -- parameters and all
    three state buffers bit-identical to the one-launch step, for one rate, the two rates of PLM-NR / the notebooks, and three."""
    import torch
    import engine as E
    import hashinit
    from schema import FULL, state_shapes
    lr, lr_bert, lr_head = rates
    cfg = E.EngineConfig(n_layers=2, trainable_layers=(0, 1), num_teachers=2)
    P = hashinit.init_state_dict(3, state_shapes(FULL, 2, cfg.D, 2))
    outs = []
    for bucketed in (False, True):
        eng = E.Engine(cfg, "cuda:0", max_batch=2, dtype="bf16")      # the bucket-wise update is the bf16 path: under fp16's dynamic
        eng.load_state_dict(P)                                          # loss scale the whole gradient is checked before any slice moves
        g = torch.Generator(device="cuda:0").manual_seed(5)
        for step in range(3):
            eng.flat_g.copy_(torch.randn(eng.n_train, device="cuda:0", generator=g) * 1e-3)
            sync = _FakeSync(eng.bucket_ranges()) if bucketed else None
            eng.step(lr, grad_scale=0.5, lr_bert=lr_bert, lr_news_head=lr_head, sync=sync)
            if bucketed:
                assert sync.waited == list(range(len(sync.ranges))) and not sync.pending      # completion order, each once
        torch.cuda.synchronize()
        outs.append([x.clone() for x in (eng.flat[True], eng.adam_m, eng.adam_v, eng.adam_vmax)] + [eng.sh[1]["w1"].clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    rng = outs[0][0]
    assert not torch.equal(rng, torch.from_numpy(np.zeros(1, np.float32)).to(rng.device).expand_as(rng))


def test_fp16_guarded_step_scans_bucket_by_bucket_and_skips_as_one():
    """fp16 under data parallelism (Engine.step(sync=...) with the dynamic loss scale's guard): every bucket is scanned for inf /
    nan behind its own collective, in completion order, and the update follows the LAST scan - a clean step is bit-identical to the
    one-launch guarded step, and one non-finite value in the bucket that arrives LAST (or in the first) leaves every slice,
    the early buckets' too, untouched and counts one skipped step."""
    import torch
    import engine as E
    import hashinit
    from schema import FULL, state_shapes
    cfg = E.EngineConfig(n_layers=2, trainable_layers=(0, 1), num_teachers=2)
    P = hashinit.init_state_dict(3, state_shapes(FULL, 2, cfg.D, 2))
    outs = []
    for bucketed in (False, True):
        eng = E.Engine(cfg, "cuda:0", max_batch=2, dtype="fp16")
        eng.load_state_dict(P)
        assert eng.scaler.enabled
        g = torch.Generator(device="cuda:0").manual_seed(5)
        ranges = eng.bucket_ranges()
        for step in range(5):
            eng.flat_g.copy_(torch.randn(eng.n_train, device="cuda:0", generator=g) * 1e-3)
            if step == 1:
                eng.flat_g[ranges[-1][1] - 1] = float("inf")          # the last element of the bucket that lands last
            if step == 3:
                eng.flat_g[ranges[0][0]] = float("nan")               # the first element of the first
            before = [x.clone() for x in (eng.flat[True], eng.adam_m, eng.adam_v, eng.adam_vmax)]
            sync = _FakeSync(ranges) if bucketed else None
            eng.step(1e-4, grad_scale=0.5, sync=sync)
            torch.cuda.synchronize()
            if bucketed:
                assert sync.waited == list(range(len(ranges))) and not sync.pending
            same = all(torch.equal(a, b) for a, b in zip(before, (eng.flat[True], eng.adam_m, eng.adam_v, eng.adam_vmax)))
            assert same == (step in (1, 3)), step
        eng.scaler.drain(eng)
        assert eng.scaler.skipped == 2 and int(eng.scaler.guard[1]) == 2 and eng.step_count == 3
        outs.append([x.clone() for x in (eng.flat[True], eng.adam_m, eng.adam_v, eng.adam_vmax)] + [eng.sh[1]["w1"].clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_native_communicator_through_the_c_abi_at_world_size_one():
    """include/tnr_hip.h tnr_comm_* (the gradient exchange behind the C ABI; replaces hvd.init / broadcast_parameters /
    DistributedOptimizer, Tiny-NewsRec/run.py:141-149): RCCL bound at run time inside libtnr_hip.so.  One GPU here, so world size 1:
    the communicator comes up on the current device, all-reduce (sum and average), reduce-scatter + all-gather and broadcast run
    asynchronously on the given stream and leave a one-rank buffer bit for bit as it was; bad arguments are refused with a message;
    and dist.GradSync(algo="native" / "native_rs_ag") drives a training step's bucketed exchange through it with the same results
    as no exchange at all.  (More than one rank needs more than one GPU: never run, like every RCCL path of this repository.)"""
    import ctypes
    import torch
    import dist as D
    import engine as E
    import hashinit
    import tnr_hip as T
    from schema import FULL, state_shapes
    L = T.lib()
    torch.cuda.set_device(0)
    ident = ctypes.create_string_buffer(128)
    assert L.tnr_comm_unique_id(ident) == 0, L.tnr_last_error()
    assert any(ident.raw)
    h = ctypes.c_void_p()
    assert L.tnr_comm_init(ident, 1, 1, ctypes.byref(h)) == -1 and b"rank" in L.tnr_last_error()        # rank out of range
    assert L.tnr_comm_init(ident, 1, 0, ctypes.byref(h)) == 0, L.tnr_last_error()
    w, r = ctypes.c_int(-1), ctypes.c_int(-1)
    assert L.tnr_comm_world(h, ctypes.byref(w), ctypes.byref(r)) == 0 and (w.value, r.value) == (1, 0)
    st = torch.cuda.Stream()
    g = torch.Generator(device="cuda:0").manual_seed(3)
    x = torch.randn(1 << 20, device="cuda:0", generator=g)
    x0 = x.clone()
    shard = torch.empty_like(x)
    torch.cuda.synchronize()
    for avg in (0, 1):
        assert L.tnr_comm_allreduce_avg(h, x.data_ptr(), x.numel(), avg, st.cuda_stream) == 0, L.tnr_last_error()
        assert L.tnr_comm_reduce_scatter_allgather(h, x.data_ptr(), shard.data_ptr(), x.numel(), avg, st.cuda_stream) == 0, L.tnr_last_error()
    assert L.tnr_comm_broadcast(h, x.data_ptr(), x.numel(), 0, st.cuda_stream) == 0, L.tnr_last_error()
    assert L.tnr_comm_broadcast(h, x.data_ptr(), x.numel(), 1, st.cuda_stream) == -1                   # root out of range
    st.synchronize()
    assert torch.equal(x, x0)
    assert L.tnr_comm_destroy(h) == 0 and L.tnr_comm_destroy(None) == 0
    # a training step's bucketed exchange through it
    cfg = E.EngineConfig(n_layers=2, trainable_layers=(0, 1), num_teachers=2)
    P = hashinit.init_state_dict(3, state_shapes(FULL, 2, cfg.D, 2))
    outs = {}
    for algo in (None, "native", "native_rs_ag"):
        eng = E.Engine(cfg, "cuda:0", max_batch=2, dtype="bf16")
        eng.load_state_dict(P)
        gs = D.GradSync(eng.flat_g, eng.bucket_ranges(), 1, force=True, algo=algo) if algo else None
        assert algo is None or (gs.native is not None and gs.algo == algo)
        gen = torch.Generator(device="cuda:0").manual_seed(5)
        for step in range(3):
            eng.flat_g.copy_(torch.randn(eng.n_train, device="cuda:0", generator=gen) * 1e-3)
            if gs is not None:
                for b in range(len(gs.ranges)):
                    gs.launch(b)
                assert len(gs.pending) == len(gs.groups)
            eng.step(1e-4, grad_scale=1.0, sync=gs)
            assert gs is None or not gs.pending
        torch.cuda.synchronize()
        outs[algo] = [t.clone() for t in (eng.flat[True], eng.adam_m, eng.adam_v, eng.adam_vmax)]
        if gs is not None:
            gs.native.close()
    for algo in ("native", "native_rs_ag"):
        for a, b in zip(outs[None], outs[algo]):
            assert torch.equal(a, b), algo
