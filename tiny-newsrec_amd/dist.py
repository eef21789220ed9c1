"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI)
replacing horovod (run.py:141-149, utils.py:43-60): flat parameter broadcast from rank 0, and a
gradient AVERAGE (hvd.Average, Compression.none, fp32) done as a few large contiguous all-reduces
(one per gradient bucket: heads first, then per trainable layer from the top its FFN block and its
attention block, Engine.bucket_ranges) launched as soon as backward has produced the bucket, so they
overlap the remaining backward GEMMs; the 1/world scale is folded into the AMSGrad kernel, which Engine.step runs bucket by
bucket in the same order, each behind its own all-reduce only (wait_bucket), so the optimiser work of the early buckets hides
under the last, exposed all-reduce.  xGMI is point-to-point, so few large messages beat many small ones."""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """-> (world, rank, local_rank); env from the launcher (torch.distributed.run sets RANK/WORLD_SIZE/LOCAL_RANK)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return world, rank, local


class GradSync:
    """Bucketed all-reduce of a flat gradient buffer. ranges: [(start, end)] in completion order.

    launch(i) is called by Engine.backward right after the last kernel that writes bucket i was enqueued on the current
    stream; the collective is stream-ordered behind it (ProcessGroupNCCL makes its own stream wait for the current one),
    so a bucket can never be reduced before it is complete.  Under the gloo backend (CPU tests, or two ranks sharing one
    GPU in the data-parallel equivalence test) device buckets are staged through host memory, synchronously."""

    def __init__(self, flat_g, ranges, world, force=False):
        self.flat_g, self.ranges, self.world = flat_g, list(ranges), world
        self.force = force and dist.is_initialized()      # exercise the collective path even with one rank
        self.pending = {}                                 # bucket -> outstanding work handle
        self.host_staged = dist.is_initialized() and dist.get_backend() == "gloo" and flat_g.is_cuda

    def launch(self, bucket):
        if self.world == 1 and not self.force:
            return
        s, e = self.ranges[bucket]
        if self.host_staged:
            h = self.flat_g[s:e].cpu()                    # synchronises with the stream that wrote the bucket
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            self.flat_g[s:e].copy_(h)
            return
        self.pending[bucket] = dist.all_reduce(self.flat_g[s:e], op=dist.ReduceOp.SUM, async_op=True)

    def wait_bucket(self, bucket):
        """Make the current stream wait for bucket's all-reduce (no-op when it was not launched asynchronously)."""
        w = self.pending.pop(bucket, None)
        if w is not None:
            w.wait()

    def wait(self):
        for b in list(self.pending):
            self.wait_bucket(b)

    @property
    def scale(self):
        return 1.0 / self.world


def broadcast_flat(tensors, src=0, force=False):
    """hvd.broadcast_parameters (run.py:142): every flat buffer from rank 0."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or force):
        for t in tensors:
            if dist.get_backend() == "gloo" and t.is_cuda:       # gloo moves host memory (tests: ranks sharing one GPU)
                h = t.cpu()
                dist.broadcast(h, src=src)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_reduce_max(x):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return x


def min_over_ranks(value):
    """Smallest `value` (an int) over the workers -- the per-epoch step count every rank can run.  The reference has no
    such guard: ranks read `sorted(files)[r::W]` (streaming.py:53), so their batch counts differ, and a rank that runs out
    first leaves the others hanging in the gradient all-reduce (dataloader.py:106-109, run.py:176)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return int(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())
