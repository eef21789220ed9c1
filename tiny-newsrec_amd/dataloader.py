"""Training loader (Tiny-NewsRec/dataloader.py:20-205): TSV lines -> news indices -> (pad, mask, label) ->
the 6-tuple run.py:175 consumes.  Bit-exact at index level with the reference (same trans_to_nindex /
pad_to_fix_len / label draw from Python's global `random`, dataloader.py:73-83,136-137).

Two delivery modes:
  * reference mode (resident=False): the 6 tensors of dataloader.py:172, gathered on the host.
  * resident mode (default on GPU): news_combined (n+1,2L) int32 and the teacher tables (T,n+1,D) fp32 are
    uploaded to HBM once; each step ships only (hist_idx (B,U) int32, mask (B,U) f32, cand_idx (B,C) int32,
    label (B,) int64) -- ~14 KB instead of ~8.1 MB -- and the gathers run in the HIP kernels.
The producer thread hands batches over a bounded queue and ends the stream with a sentinel (the reference's
`aval_count` end-of-stream test races)."""
import logging
import queue
import random
import threading

import numpy as np
import torch

from decode_worker import decode_lines, decode_process, start_without_main
from streaming import StreamSampler, _count_lines, shard_files

_END = object()


class IndexBatch(tuple):
    """(hist_idx, log_mask, cand_idx, label[, dedup plan or None]) in resident mode."""


class RowBatch(tuple):
    """The reference's 6-tuple (dataloader.py:172) whose tensors came over in one staged copy (ready = its event)."""


class DataLoaderTrain:
    def __init__(self, data_dir, filename_pat, args, world_size, worker_rank, cuda_device_idx, news_index,
                 news_combined, teacher_embs, word_dict=None, enable_prefetch=True, enable_shuffle=False,
                 enable_gpu=True, resident=None):
        self.data_dir, self.filename_pat = data_dir, filename_pat
        self.npratio, self.user_log_length, self.batch_size = args.npratio, args.user_log_length, args.batch_size
        self.worker_rank, self.world_size, self.cuda_device_idx = worker_rank, world_size, cuda_device_idx
        self.shuffle_buffer_size = args.shuffle_buffer_size
        self.enable_prefetch, self.enable_shuffle, self.enable_gpu = enable_prefetch, enable_shuffle, enable_gpu
        self.num_teachers = args.num_teachers
        self.teacher_embs, self.news_combined, self.news_index = teacher_embs, news_combined, news_index
        self.resident = enable_gpu if resident is None else resident
        self.dedup = bool(getattr(args, "dedup_news", False))
        self.decode_process = bool(getattr(args, "decode_process", False)) and enable_prefetch
        self.epoch = -1
        self.sampler = None
        self.dev_tables = None
        self._thread = None
        if self.resident and enable_gpu:
            dev = torch.device("cuda", cuda_device_idx)
            self.dev_news = torch.from_numpy(np.ascontiguousarray(news_combined, dtype=np.int32)).to(dev)
            if self.num_teachers:
                self.dev_tables = torch.from_numpy(np.stack([np.asarray(t, np.float32) for t in teacher_embs], 0)).to(dev)

    # -- index level (bit-exact with the reference) ------------------------------------------------
    def trans_to_nindex(self, nids):
        ni = self.news_index
        return [ni[i] if i in ni else 0 for i in nids]

    def pad_to_fix_len(self, x, fix_length, padding_front=True, padding_value=0):
        n = len(x)
        if padding_front:
            return [padding_value] * (fix_length - n) + x[-fix_length:], [0] * (fix_length - n) + [1] * min(fix_length, n)
        return x[-fix_length:] + [padding_value] * (fix_length - n), [1] * min(fix_length, n) + [0] * (fix_length - n)

    def decode(self, batch):
        """lines -> hist_idx (B,U), mask (B,U), cand_idx (B,C), label (B,)   dataloader.py:119-149 (decode_worker.decode_lines)"""
        return decode_lines(batch, self.news_index, self.user_log_length, self.npratio)

    def _staging(self, words):
        """A pinned host buffer of >= `words` int32 from a small ring (the H2D copy of the batch that last used it has completed)."""
        if not hasattr(self, "_pin"):
            self._pin, self._pin_i = [None] * 16, 0
        i = self._pin_i = (self._pin_i + 1) % len(self._pin)
        slot = self._pin[i]
        if slot is not None:
            slot[1].synchronize()
        if slot is None or slot[0].numel() < words:
            slot = [torch.empty(max(words, 8192), dtype=torch.int32).pin_memory(), None]
        self._pin[i] = slot
        return slot

    def _process(self, batch):
        h, m, c, y = self.decode(batch)
        if self.resident and self.enable_gpu:
            plan = None
            if self.dedup:
                from dedup import build_plan
                plan = build_plan(h, c)
            return self._to_device(h, m, c, y, plan)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x))
        out = [t(self.news_combined[h].astype(np.int64)), t(m), t(self.news_combined[c].astype(np.int64)), t(y),
               [t(np.asarray(te)[h].astype(np.float32)) for te in self.teacher_embs[:self.num_teachers]],
               [t(np.asarray(te)[c].astype(np.float32)) for te in self.teacher_embs[:self.num_teachers]]]
        if self.enable_gpu and self.enable_prefetch and threading.current_thread() is self._thread:
            return self._rows_to_device(out)                    # producer thread: one staged copy on its own stream
        if self.enable_gpu:
            out = [x.cuda() if isinstance(x, torch.Tensor) else [v.cuda() for v in x] for x in out]
        return tuple(out)

    def _rows_to_device(self, out):
        """Reference mode from the producer thread: the reference's 4 + 2 T `.cuda()` calls (dataloader.py:160-170) are pageable copies
        on the default stream - 8.1 MB per step that queue between the training thread's kernels (+0.8 ms on a 7.6 ms step).  Here the
        same tensors leave as ONE pinned staging buffer and ONE asynchronous copy on the producer's own stream; the consumer's stream
        waits for its event (__next__).  Same 6-tuple, same values."""
        dev = torch.device("cuda", self.cuda_device_idx)
        flat = [out[0], out[1], out[2], out[3]] + list(out[4]) + list(out[5])
        words = [x.numpy().reshape(-1).view(np.int32) for x in flat]       # int64 parts as pairs of words
        off, offs = 0, []
        for a in words:
            off += off & 1                                      # 8-byte alignment for every part
            offs.append((off, a.size))
            off += a.size
        slot = self._staging(off)
        host = slot[0].numpy()
        for a, (o, n) in zip(words, offs):
            host[o:o + n] = a
        if not hasattr(self, "_copy_stream"):
            self._copy_stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(self._copy_stream):
            d = slot[0][:off].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        slot[1] = ev
        dv = [d[o:o + n].view(x.dtype).view(x.shape) for x, (o, n) in zip(flat, offs)]
        T_ = len(out[4])
        res = RowBatch((dv[0], dv[1], dv[2], dv[3], dv[4:4 + T_], dv[4 + T_:]))
        res.ready, res.buf = ev, d
        return res

    def _to_device(self, h, m, c, y, plan):
        """Resident mode: ONE pinned staging buffer and ONE asynchronous copy per batch on the producer's own stream (nine small
        pageable copies cost the producer ~0.4 ms of a 2.8 ms step); the consumer's stream waits for the copy's event (__next__)."""
        dev = self.dev_news.device
        B, U, Cn = h.shape[0], h.shape[1], c.shape[1]
        parts = [("h", np.ascontiguousarray(h, np.int32).reshape(-1)), ("c", np.ascontiguousarray(c, np.int32).reshape(-1)),
                 ("m", np.ascontiguousarray(m, np.float32).reshape(-1).view(np.int32)),
                 ("y", np.ascontiguousarray(y, np.int64).view(np.int32))]     # int64 labels as pairs of words (offsets kept even below)
        if plan is not None:
            parts += [(k, getattr(plan, k)) for k in ("uniq", "inv", "order", "seg")]
        off, offs = 0, {}
        for k, a in parts:
            off += off & 1                                      # 8-byte alignment for every part
            offs[k] = (off, a.size)
            off += a.size
        slot = self._staging(off)
        host = slot[0].numpy()
        for k, a in parts:
            o, n = offs[k]
            host[o:o + n] = a
        if not hasattr(self, "_copy_stream"):
            self._copy_stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(self._copy_stream):
            d = slot[0][:off].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        slot[1] = ev
        part = lambda k: d[offs[k][0]:offs[k][0] + offs[k][1]]
        dplan = None
        if plan is not None:
            from dedup import DedupPlan
            dplan = DedupPlan()
            for k in ("uniq", "inv", "order", "seg"):
                setattr(dplan, k, part(k))
            dplan.n_enc, dplan.n_unique, dplan.n_slots = plan.n_enc, plan.n_unique, plan.n_slots
        out = IndexBatch((part("h").view(B, U), part("m").view(torch.float32).view(B, U), part("c").view(B, Cn),
                          part("y").view(torch.int64), dplan))
        out.ready, out.buf = ev, d
        return out

    # -- streaming -----------------------------------------------------------------------------------
    def _sampler_args(self):
        return dict(data_dir=self.data_dir, filename_pat=self.filename_pat, batch_size=self.batch_size,
                    worker_rank=self.worker_rank, world_size=self.world_size, enable_shuffle=self.enable_shuffle,
                    shuffle_buffer_size=self.shuffle_buffer_size, shuffle_seed=self.epoch)

    def _produce_from_process(self):
        """Resident mode with --decode_process: the epoch's sampler + decode + de-duplication plan run in a CHILD process
        (spawned from decode_worker.py, which imports neither torch nor this module: it never touches the GPU and is up in a
        fraction of a second), this thread only stages its arrays and starts the H2D copy.  The decoder then no longer
        shares the GIL with the thread that launches the kernels - what held run.py's default mode to 0.87 of bench.py's figure.
        The label draws stay where the reference has them: the child continues Python's global `random` stream from this process's
        state and hands the state back at the end of the epoch."""
        import multiprocessing as mp
        from dedup import DedupPlan
        self.epoch += 1
        ctx = mp.get_context("spawn")
        q = ctx.Queue(32)
        cfg = dict(sampler=self._sampler_args(), news_index=self.news_index, user_log_length=self.user_log_length,
                   npratio=self.npratio, dedup=self.dedup)
        proc = ctx.Process(target=decode_process, args=(cfg, random.getstate(), q), daemon=True)
        start_without_main(proc)             # no re-import of run.py (and with it torch) in the child
        self._proc = proc
        try:
            while True:
                try:
                    item = q.get(timeout=1.0)
                except queue.Empty:
                    if self.stopped:
                        return
                    if not proc.is_alive():
                        raise RuntimeError("the decode process died (exit code %s)" % proc.exitcode)
                    continue
                if item[0] == "end":
                    random.setstate(item[1])
                    return
                if item[0] == "error":
                    raise RuntimeError("decode process: " + item[1])
                if self.stopped:
                    return
                h, m, c, y, pl = item[1]
                plan = None
                if pl is not None:
                    plan = DedupPlan()
                    plan.uniq, plan.inv, plan.order, plan.seg, plan.n_enc, plan.n_unique, plan.n_slots = pl
                self.outputs.put(self._to_device(h, m, c, y, plan))
        finally:
            if proc.is_alive():
                proc.terminate()
            proc.join(5)
            self._proc = None

    def _new_sampler(self):
        self.epoch += 1
        self.sampler = StreamSampler(data_dir=self.data_dir, filename_pat=self.filename_pat, batch_size=self.batch_size,
                                     worker_rank=self.worker_rank, world_size=self.world_size,
                                     enable_shuffle=self.enable_shuffle, shuffle_buffer_size=self.shuffle_buffer_size,
                                     shuffle_seed=self.epoch)   # epoch id as shuffle seed (dataloader.py:69)
        return iter(self.sampler)

    def next_epoch_batches(self, stat=None):
        """Number of batches the NEXT iter(self) yields on this worker: the sampler it builds re-shards the files with the
        epoch number as the shuffle seed (dataloader.py:61-70 of the reference), so a worker's file set -- and with unequal
        files its batch count -- changes from epoch to epoch.  run.py all-reduces the minimum of this before every epoch
        (dist.min_over_ranks): a rank that ran out of batches earlier than the others would leave them waiting in the
        gradient all-reduce.  stat: {path: lines} (streaming.get_stat) or None to count here."""
        files = shard_files(self.data_dir, self.worker_rank, self.world_size, self.filename_pat, self.enable_shuffle,
                            self.epoch + 1)
        n = sum(stat[f] if stat is not None and f in stat else _count_lines(f) for f in files)
        return -(-n // self.batch_size)

    def _produce(self):
        try:
            if self.enable_gpu:
                torch.cuda.set_device(self.cuda_device_idx)      # dataloader.py:86-88
            if self.decode_process and self.resident and self.enable_gpu:
                self._produce_from_process()
            else:
                for batch in self._new_sampler():
                    if self.stopped:
                        break
                    self.outputs.put(self._process(batch))
        except BaseException as e:      # surface producer failures instead of hanging the consumer
            logging.exception("producer failed")
            self.outputs.put(e)
            return
        self.outputs.put(_END)

    def __iter__(self):
        self.join()
        self.stopped = False
        if self.enable_prefetch:
            self.outputs = queue.Queue(10)
            self._thread = threading.Thread(target=self._produce, daemon=True)
            self._thread.start()
        else:
            self._sync_it = self._new_sampler()
        return self

    def __next__(self):
        if self.enable_prefetch:
            item = self.outputs.get()
            if item is _END:
                raise StopIteration
            if isinstance(item, BaseException):
                raise item
        else:
            item = self._process(next(self._sync_it))
        ev = getattr(item, "ready", None)
        if ev is not None:                       # resident mode: the batch's one H2D copy runs on the producer's stream
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            item.buf.record_stream(cur)          # allocated on the copy stream, read by this one
        return item

    def join(self):
        self.stopped = True
        if self._thread is not None:
            while self._thread.is_alive():
                try:
                    self.outputs.get(timeout=0.05)
                except queue.Empty:
                    pass
            self._thread = None
        self.sampler = None


class DataLoaderTest(DataLoaderTrain):
    """Eval loader (Tiny-NewsRec/dataloader.py:208-314) at index level: lines `iid uid time history impressions`
    with impressions = "N1-0 N2-1 ..." -> (hist_idx (B,U) int32, mask (B,U) f32, [cand_idx arrays], [label arrays])."""

    def __init__(self, data_dir, filename_pat, args, world_size, worker_rank, cuda_device_idx, news_index, news_scoring=None,
                 word_dict=None, enable_prefetch=True, enable_shuffle=False, enable_gpu=True):
        super().__init__(data_dir, filename_pat, args, world_size, worker_rank, cuda_device_idx, news_index, None, [],
                         word_dict, enable_prefetch, enable_shuffle, enable_gpu, resident=False)
        self.news_scoring = news_scoring

    def _process(self, batch):
        H, M, C, Y = [], [], [], []
        for raw in batch:
            line = raw.decode("utf-8").split("\t")
            click, mask = self.pad_to_fix_len(self.trans_to_nindex(line[3].split()), self.user_log_length)
            imp = line[4].split()
            H.append(click)
            M.append(mask)
            C.append(np.asarray(self.trans_to_nindex([i.split("-")[0] for i in imp]), np.int64))
            Y.append(np.asarray([int(i.split("-")[1]) for i in imp]))
        h, m = torch.from_numpy(np.asarray(H, np.int32)), torch.from_numpy(np.asarray(M, np.float32))
        if self.enable_gpu:
            h, m = h.cuda(), m.cuda()
        return h, m, C, Y
