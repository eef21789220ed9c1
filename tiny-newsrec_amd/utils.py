"""Host helpers mirroring Tiny-NewsRec/utils.py for the training path: str2bool (:32-40), worker bootstrap
(init_hvd_cuda :43-60, here over torch.distributed/RCCL), logger format (:63-70), acc (:79-83), checkpoint
discovery (:127-145).  The reference's unused GloVe loader and the eval metrics are out of scope."""
import argparse
import logging
import os
import sys

import torch


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def init_hvd_cuda(enable_hvd=True, enable_gpu=True):
    """-> (size, rank, local_rank).  One process per GPU; rank/world come from the launcher's env
    (torch.distributed.run).  Name kept for run.py compatibility -- there is no horovod underneath."""
    size, rank, local = 1, 0, 0
    if enable_hvd:
        import dist
        size, rank, local = dist.init("nccl" if enable_gpu else "gloo")
        logging.info(f"world_size:{size}, rank:{rank}, local_rank:{local}")
    if enable_gpu:
        torch.cuda.set_device(local)
    return size, rank, local


def setuplogger():
    root = logging.getLogger()
    root.setLevel(logging.INFO)
    h = logging.StreamHandler(sys.stdout)
    h.setLevel(logging.INFO)
    h.setFormatter(logging.Formatter("[%(levelname)s %(asctime)s] %(message)s"))
    root.addHandler(h)


def acc(y_true, y_hat):
    """mean(argmax(score) == y)   utils.py:79-83"""
    hit = torch.sum(y_true == torch.argmax(y_hat, dim=-1))
    return hit.data.float() * 1.0 / y_true.shape[0]


def latest_checkpoint(directory):
    """epoch-N.pt with the largest N (utils.py:127-137)."""
    if not os.path.exists(directory):
        return None
    found = {}
    for x in os.listdir(directory):
        try:
            found[int(x.split(".")[-2].split("-")[-1])] = x
        except (ValueError, IndexError):
            continue
    return os.path.join(directory, found[max(found)]) if found else None


def get_checkpoint(directory, ckpt_name):
    p = os.path.join(directory, ckpt_name)
    return p if os.path.exists(p) else None
