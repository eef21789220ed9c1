"""The line decode of the training loader (Tiny-NewsRec/dataloader.py:119-149) as a free function, and the child process that
runs it for DataLoaderTrain's resident mode.  Imports numpy and the standard library only: the spawned child must not pay for
(or touch) torch."""
import random
import threading

import numpy as np


def decode_lines(batch, news_index, U, K):
    """lines -> hist_idx (B,U), mask (B,U), cand_idx (B,K+1), label (B,)
    The same values as trans_to_nindex / pad_to_fix_len / the label draw of the reference (dataloader.py:73-83, 131-137), line by
    line and in its order (one random.randint per line from Python's global generator), written into preallocated arrays: in
    the producer THREAD this code shares the GIL with the thread that launches the kernels, so its Python time per batch is what
    bounds a file-fed run."""
    B = len(batch)
    H, M = np.zeros((B, U), np.int64), np.zeros((B, U), np.float32)
    C, Y = np.empty((B, K + 1), np.int64), np.empty(B, np.int64)
    get, randint = news_index.get, random.randint
    for r, raw in enumerate(batch):
        line = raw.decode("utf-8").split("\t")
        click = [get(i, 0) for i in line[3].split()]           # unknown id -> index 0 (its mask stays 1)
        n = len(click)
        if n >= U:
            H[r] = click[-U:]                                  # the LAST U clicks
            M[r] = 1.0
        elif n:
            H[r, U - n:] = click                               # left-padded with 0
            M[r, U - n:] = 1.0
        pos = [get(i, 0) for i in line[4].split()]
        neg = [get(i, 0) for i in line[5].split()]
        label = randint(0, K)
        C[r] = neg[:label] + pos + neg[label:]
        Y[r] = label
    return H, M, C, Y


def decode_process(cfg, rnd_state, q):
    """Child process of DataLoaderTrain._produce_from_process: one epoch's lines -> index arrays (+ de-duplication plan).
    Continues Python's global `random` stream from the parent's state and hands the state back with the end marker."""
    try:
        from streaming import StreamSampler
        random.setstate(rnd_state)
        build_plan = None
        if cfg["dedup"]:
            from dedup import build_plan
        for batch in StreamSampler(**cfg["sampler"]):
            h, m, c, y = decode_lines(batch, cfg["news_index"], cfg["user_log_length"], cfg["npratio"])
            pl = None
            if build_plan is not None:
                p = build_plan(h, c)
                pl = None if p is None else (p.uniq, p.inv, p.order, p.seg, p.n_enc, p.n_unique, p.n_slots)
            q.put(("batch", (h.astype(np.int32), m, c.astype(np.int32), y, pl)))
        q.put(("end", random.getstate()))
    except BaseException as e:          # noqa: BLE001 - reported to the parent, which raises
        import traceback
        q.put(("error", "%r\n%s" % (e, traceback.format_exc())))


_SPAWN_LOCK = threading.Lock()       # start_without_main patches __main__, which is process-global: one spawn at a time


def start_without_main(proc):
    """proc.start() of a "spawn" process WITHOUT the child re-importing the parent's __main__ (multiprocessing does that for every
    spawned child: run.py's top-level `import torch` would then cost the decoder a second or two at the start of every epoch and
    put a GPU framework into a process that is there to stay off the GPU).  The child's target lives in this module and needs
    nothing from __main__.  The patch of __main__ is process-global: every spawn of this package goes through here under one
    lock, so two loaders (train + eval) starting their decoders at the same moment cannot restore each other's half-patched
    state, and the window in which __main__.__file__ is missing is the few hundred microseconds of Popen itself."""
    import sys
    with _SPAWN_LOCK:
        main = sys.modules.get("__main__")
        saved = {}
        for k in ("__spec__", "__file__"):
            if main is not None and hasattr(main, k):
                saved[k] = getattr(main, k)
        try:
            if main is not None:
                if "__spec__" in saved:
                    main.__spec__ = None
                if "__file__" in saved:
                    del main.__file__
            proc.start()
        finally:
            for k, v in saved.items():
                setattr(main, k, v)


def report_modules(q, names):
    """test hook: which of `names` the spawned child has imported"""
    import sys
    q.put([n for n in names if sys.modules.get(n) is not None])
