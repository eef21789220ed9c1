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


class NativeComm:
    """The C-ABI communicator (include/tnr_hip.h tnr_comm_*: RCCL bound at run time inside libtnr_hip.so) on a stream of its own.
    The 128-byte unique id goes from rank 0 to the others through the process group torch.distributed already has (any
    backend: it is host bytes), or nowhere at world size 1."""

    def __init__(self, world, rank, device):
        import ctypes
        import tnr_hip as T
        self.T, self.world, self.rank = T, world, rank
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            rc = T.lib().tnr_comm_unique_id(ident)
            if rc != 0:
                raise T.TnrError("tnr_comm_unique_id failed (%d): %s" % (rc, T.lib().tnr_last_error().decode()))
        if world > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0)
            ident = ctypes.create_string_buffer(box[0], 128)
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            rc = T.lib().tnr_comm_init(ident, world, rank, ctypes.byref(handle))
        if rc != 0:
            raise T.TnrError("tnr_comm_init failed (%d): %s" % (rc, T.lib().tnr_last_error().decode()))
        self.handle = handle
        self.stream = torch.cuda.Stream(device)

    def _run(self, name, *args):
        """`name`(comm, *args, comm stream) behind everything enqueued on the current stream so far -> an event-backed work handle."""
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        rc = getattr(self.T.lib(), name)(self.handle, *args, self.stream.cuda_stream)
        if rc != 0:
            raise self.T.TnrError("%s failed (%d): %s" % (name, rc, self.T.lib().tnr_last_error().decode()))
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return _EventWork(ev)

    def allreduce_sum(self, t):
        return self._run("tnr_comm_allreduce_avg", t.data_ptr(), t.numel(), 0)

    def reduce_scatter_allgather_sum(self, t, shard):
        return self._run("tnr_comm_reduce_scatter_allgather", t.data_ptr(), shard.data_ptr(), t.numel(), 0)

    def broadcast(self, t, src=0):
        self._run("tnr_comm_broadcast", t.data_ptr(), t.numel(), src).wait()

    def close(self):
        if self.handle is not None:
            self.stream.synchronize()
            self.T.lib().tnr_comm_destroy(self.handle)
            self.handle = None


class _EventWork:
    """What GradSync needs of a work handle: wait() puts the CURRENT stream behind the collective."""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class GradSync:
    """Bucketed all-reduce of a flat gradient buffer. ranges: [(start, end)] in completion order.

    launch(i) is called by Engine.backward right after the last kernel that writes bucket i was enqueued on the current
    stream; the collective is stream-ordered behind it (ProcessGroupNCCL makes its own stream wait for the current one),
    so a bucket can never be reduced before it is complete.  It runs ASYNCHRONOUSLY (pending[i] holds its work handle) and
    wait_bucket(i) puts the current stream behind it -- Engine.step(sync=...) calls that bucket by bucket, so each AMSGrad slice
    waits for its own collective only.
    Under the gloo backend (CPU tests, or two ranks sharing one GPU in the data-parallel equivalence tests) a device bucket is
    staged through a pinned host mirror: launch() copies it out (the host waits for that copy -- gloo needs the bytes) and starts
    the all-reduce asynchronously on the mirror; wait_bucket() waits for it and copies the sum back on the current stream.  The
    bookkeeping (pending / wait_bucket order / per-bucket optimiser slices behind in-flight collectives) is the same code path
    RCCL takes.
    timing = True: wait_bucket brackets its wait with events on the current stream; exposed_ms() = the time the compute stream
    actually stood still behind collectives (what overlap did not hide), per bucket, since the last call."""

    def __init__(self, flat_g, ranges, world, force=False, timing=False, merge_layers=False, algo="allreduce"):
        """merge_layers: one collective per trainable layer (its FFN and attention buckets together: 3 collectives of 2.8 / 28 /
        28 MB instead of 5 for the headline model) - fewer, larger messages; the layer's collective then starts when its
        attention block is complete.  algo: "allreduce" (one all-reduce per collective; RCCL picks ring / tree by NCCL_ALGO) or
        "rs_ag" (reduce-scatter into this rank's 1 / world shard, then all-gather: on a fully connected xGMI mesh every rank
        exchanges directly with every other; nccl backend only, falls back to all-reduce elsewhere).  Both are knobs for the
        first multi-GPU run (bench.py --dp-sweep); results are the same sums."""
        self.flat_g, self.ranges, self.world = flat_g, list(ranges), world
        self.force = force and (dist.is_initialized() or algo in ("native", "native_rs_ag"))   # the collective path even with one rank
        self.pending = {}                                 # collective -> (work handle, pinned host mirror or None)
        self.host_staged = dist.is_initialized() and dist.get_backend() == "gloo" and flat_g.is_cuda
        self._mirror = {}
        self.timing = timing and flat_g.is_cuda
        self._events = []                                 # (bucket, start event, end event)
        # buckets (the engine's units of completion) -> collectives
        n = len(self.ranges)
        if merge_layers and n >= 3 and (n - 1) % 2 == 0:
            self.groups = [[0]] + [[b, b + 1] for b in range(1, n, 2)]
        else:
            self.groups = [[b] for b in range(n)]
        self.group_of = {b: g for g, mem in enumerate(self.groups) for b in mem}
        self.group_range = []
        for mem in self.groups:
            s_, e_ = min(self.ranges[b][0] for b in mem), max(self.ranges[b][1] for b in mem)
            assert sum(self.ranges[b][1] - self.ranges[b][0] for b in mem) == e_ - s_, "merged gradient buckets must be adjacent"
            self.group_range.append((s_, e_))
        self._arrived = {}
        self.algo = algo if (dist.is_initialized() and dist.get_backend() == "nccl") else "allreduce"
        self._shard = {}
        # algo "native" / "native_rs_ag": the same collectives through the library's own C ABI (tnr_comm_*) instead of
        # ProcessGroupNCCL; every in-flight collective of a size gets a shard of its own (the communicator's one stream orders them,
        # as ProcessGroupNCCL's does, but nothing is assumed about it here)
        self.native = None
        if algo in ("native", "native_rs_ag") and flat_g.is_cuda and (world > 1 or force):
            self.native = NativeComm(world, dist.get_rank() if dist.is_initialized() else 0, flat_g.device)
            self.algo = algo
            self.host_staged = False

    def bucket_bytes(self):
        return [4 * (e - s) for s, e in self.ranges]

    def collective_bytes(self):
        return [4 * (e - s) for s, e in self.group_range]

    def _collective(self, t):
        """sum over ranks of the flat slice t, in place, asynchronously -> work handle"""
        if self.native is not None:
            if self.algo == "native_rs_ag" and t.numel() % max(self.world, 1) == 0:
                key = (t.data_ptr(), t.numel())
                shard = self._shard.get(key)
                if shard is None:
                    shard = self._shard[key] = torch.empty(t.numel() // max(self.world, 1), dtype=t.dtype, device=t.device)
                return self.native.reduce_scatter_allgather_sum(t, shard)
            return self.native.allreduce_sum(t)
        if self.algo == "rs_ag" and t.numel() % max(self.world, 1) == 0:
            w = max(self.world, 1)
            k = t.numel() // w
            # ONE shard buffer per size, shared by every in-flight collective of that size (the two per-layer buckets of a model
            # have equal sizes).  Safe only because ProcessGroupNCCL issues all collectives of this group on ONE internal stream in
            # call order: the reduce-scatter of collective n + 1 cannot start before the all-gather of collective n has read the
            # shard.  A backend with one stream per collective would need a shard per in-flight collective.
            shard = self._shard.get(k)
            if shard is None:
                shard = self._shard[k] = torch.empty(k, dtype=t.dtype, device=t.device)
            dist.reduce_scatter_tensor(shard, t, op=dist.ReduceOp.SUM, async_op=True)
            return dist.all_gather_into_tensor(t, shard, async_op=True)      # same communicator stream: ordered behind the scatter
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    def launch(self, bucket):
        if self.world == 1 and not self.force:
            return
        g = self.group_of[bucket]
        self._arrived[g] = self._arrived.get(g, 0) + 1
        if self._arrived[g] < len(self.groups[g]):
            return                                        # the layer's other bucket is still being written
        self._arrived[g] = 0
        bucket = g
        s, e = self.group_range[g]
        if self.host_staged:
            h = self._mirror.get(bucket)
            if h is None or h.numel() != e - s:
                h = self._mirror[bucket] = torch.empty(e - s, dtype=self.flat_g.dtype, pin_memory=True)
            h.copy_(self.flat_g[s:e], non_blocking=True)  # stream-ordered behind the kernels that wrote the bucket
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()
            self.pending[bucket] = (dist.all_reduce(h, op=dist.ReduceOp.SUM, async_op=True), h)
            return
        self.pending[bucket] = (self._collective(self.flat_g[s:e]), None)

    def wait_bucket(self, bucket):
        """Make the current stream wait for the collective that carries `bucket` (no-op when it was not launched or has been
        waited for already through another bucket of the same collective)."""
        bucket = self.group_of[bucket]
        w = self.pending.pop(bucket, None)
        if w is None:
            return
        work, h = w
        if self.timing:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        work.wait()
        if h is not None:
            s, e = self.group_range[bucket]
            self.flat_g[s:e].copy_(h, non_blocking=True)
        if self.timing:
            e1.record()
            self._events.append((bucket, e0, e1))

    def wait(self):
        for g in list(self.pending):
            self.wait_bucket(self.groups[g][0])

    def exposed_ms(self):
        """-> {collective: [ms, ...]} of the waits recorded since the last call (synchronises the device).  Keys are COLLECTIVE
        indices (self.groups / collective_bytes order): with merge_layers a layer's two buckets share one."""
        torch.cuda.synchronize()
        out = {}
        for b, e0, e1 in self._events:
            out.setdefault(b, []).append(e0.elapsed_time(e1))
        self._events = []
        return out

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
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])     # this rank's GPU, not a guess from the rank number
        else:
            dist.barrier()


def all_reduce_max(x):
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and x.is_cuda:           # gloo moves host memory (tests: ranks sharing one GPU)
            h = x.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            x.copy_(h)
        else:
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
