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

from streaming import StreamSampler, _count_lines, shard_files

_END = object()


class IndexBatch(tuple):
    """(hist_idx, log_mask, cand_idx, label[, dedup plan or None]) in resident mode."""


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
        """lines -> hist_idx (B,U), mask (B,U), cand_idx (B,C), label (B,)   dataloader.py:119-149"""
        H, M, C, Y = [], [], [], []
        for raw in batch:
            line = raw.decode("utf-8").split("\t")
            click, mask = self.pad_to_fix_len(self.trans_to_nindex(line[3].split()), self.user_log_length)
            pos = self.trans_to_nindex(line[4].split())
            neg = self.trans_to_nindex(line[5].split())
            label = random.randint(0, self.npratio)
            H.append(click)
            M.append(mask)
            C.append(neg[:label] + pos + neg[label:])
            Y.append(label)
        return (np.asarray(H, np.int64), np.asarray(M, np.float32), np.asarray(C, np.int64), np.asarray(Y, np.int64))

    def _process(self, batch):
        h, m, c, y = self.decode(batch)
        if self.resident and self.enable_gpu:
            dev = self.dev_news.device
            plan = None
            if self.dedup:
                from dedup import build_plan
                plan = build_plan(h, c)
                plan = plan.to(dev) if plan is not None else None
            return IndexBatch((torch.from_numpy(h.astype(np.int32)).to(dev, non_blocking=True), torch.from_numpy(m).to(dev),
                               torch.from_numpy(c.astype(np.int32)).to(dev), torch.from_numpy(y).to(dev), plan))
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x))
        out = [t(self.news_combined[h].astype(np.int64)), t(m), t(self.news_combined[c].astype(np.int64)), t(y),
               [t(np.asarray(te)[h].astype(np.float32)) for te in self.teacher_embs[:self.num_teachers]],
               [t(np.asarray(te)[c].astype(np.float32)) for te in self.teacher_embs[:self.num_teachers]]]
        if self.enable_gpu:
            out = [x.cuda() if isinstance(x, torch.Tensor) else [v.cuda() for v in x] for x in out]
        return tuple(out)

    # -- streaming -----------------------------------------------------------------------------------
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
            return item
        return self._process(next(self._sync_it))

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
