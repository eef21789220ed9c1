"""TF-free restatement of Tiny-NewsRec/streaming.py: the data-parallel file-sharding rule (get_worker_files
:40-58, bit-exact incl. the seeded shuffle) and a line reader with the semantics the reference asks tf.data for
(:61-96): round-robin interleave over this worker's files in blocks of 128 lines, optional shuffle buffer,
batches of `batch_size` raw byte lines, last batch may be short (no drop_remainder, :76)."""
import fnmatch
import logging
import os
import random

import numpy as np


def _matching(dirname, pattern, recursive=False):
    """Paths under dirname whose base name matches the glob `pattern` (None when the directory is missing)."""
    if not os.path.exists(dirname):
        logging.warning(f"{dirname} does not exist!")
        return None
    hits = []
    for entry in os.listdir(dirname):
        full = os.path.join(dirname, entry)
        if os.path.isdir(full):
            if recursive:
                hits += _matching(full, pattern) or []
        elif fnmatch.fnmatch(entry, pattern):
            hits.append(full)
    return hits


def _count_lines(path):
    with open(path, "rb") as f:
        return sum(block.count(b"\n") for block in iter(lambda: f.read(1 << 20), b""))


def get_stat(dirname, filename_pat="*"):
    """{path: number of lines} (streaming.py:10-22 shells out to `wc -l`; counted here directly)."""
    paths = _matching(dirname, filename_pat)
    return None if paths is None else {p: _count_lines(p) for p in paths}


def get_files(dirname, filename_pat="*", recursive=False):
    return _matching(dirname, filename_pat, recursive)


def get_worker_files(dirname, worker_rank, world_size, filename_pat="*", shuffle=False, seed=0):
    """The data-parallel sharding rule (streaming.py:40-58), bit-exact: sort the matches, optionally
    random.seed(seed) + random.shuffle, then this worker takes every world_size-th file starting at its rank.
    NB the seeding is a side effect the label draw of dataloader.py:136 inherits, exactly as in the reference."""
    ordered = sorted(_matching(dirname, filename_pat))
    if shuffle:
        random.seed(seed)
        random.shuffle(ordered)
    mine = ordered[worker_rank::world_size]
    logging.info("worker_rank:%s, world_size:%s, shuffle:%s, seed:%s, directory:%s, files:%s",
                 worker_rank, world_size, shuffle, seed, dirname, mine)
    return mine


def shard_files(dirname, worker_rank, world_size, filename_pat="*", shuffle=False, seed=0):
    """The same file list as get_worker_files WITHOUT its side effect on Python's global `random` (and without its log
    line): random.shuffle under random.seed(seed) is random.Random(seed).shuffle.  For asking "which files will epoch e
    give this worker" from the main thread while the producer thread owns the global generator."""
    ordered = sorted(_matching(dirname, filename_pat))
    if shuffle:
        random.Random(seed).shuffle(ordered)
    return ordered[worker_rank::world_size]


class StreamReader:
    BLOCK = 128

    def __init__(self, data_paths, batch_size, shuffle=False, shuffle_buffer_size=1000, seed=None):
        self.paths, self.batch_size = list(data_paths), batch_size
        self.shuffle, self.buf_size = shuffle, shuffle_buffer_size
        self.rng = random.Random(seed)
        self.endofstream = True
        self._it = None

    def _lines(self):
        files = [open(p, "rb") for p in self.paths]
        try:
            live = list(files)
            while live:
                for f in list(live):
                    for _ in range(self.BLOCK):
                        line = f.readline()
                        if not line:
                            live.remove(f)
                            break
                        yield line.rstrip(b"\n")
        finally:
            for f in files:
                f.close()

    def _shuffled(self):
        buf = []
        for line in self._lines():
            if len(buf) < self.buf_size:
                buf.append(line)
                continue
            i = self.rng.randrange(len(buf))
            buf[i], line = line, buf[i]
            yield line
        self.rng.shuffle(buf)
        yield from buf

    def reset(self):
        self._it = self._shuffled() if self.shuffle else self._lines()
        self.endofstream = False

    def get_next(self):
        batch = []
        for line in self._it:
            batch.append(line)
            if len(batch) == self.batch_size:
                break
        if not batch:
            self.endofstream = True
            return None
        return np.array(batch, dtype=object)

    def reach_end(self):
        return self.endofstream


class StreamSampler:
    def __init__(self, data_dir, filename_pat, batch_size, worker_rank, world_size, enable_shuffle=False,
                 shuffle_buffer_size=1000, shuffle_seed=0):
        paths = get_worker_files(data_dir, worker_rank, world_size, filename_pat, shuffle=enable_shuffle, seed=shuffle_seed)
        self.stream_reader = StreamReader(paths, batch_size, enable_shuffle, shuffle_buffer_size)

    def __iter__(self):
        self.stream_reader.reset()
        return self

    def __next__(self):
        nb = self.stream_reader.get_next()
        if nb is None:
            raise StopIteration
        return nb

    def reach_end(self):
        return self.stream_reader.reach_end()
