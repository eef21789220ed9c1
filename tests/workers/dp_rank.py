"""One data-parallel worker of tests/test_dp_gpu.py (started once per rank, all on cuda:0, torch.distributed over gloo so
that two ranks can share one GPU; device buckets are staged through host memory by dist.GradSync).

Each rank takes impressions [rank * B/W, (rank+1) * B/W) of a B-impression batch, runs Engine.forward / backward with the
bucketed all-reduce launched from the backward's after_bucket hook exactly as run.py / bench.py do, steps AMSGrad with the
1/W scale, and dumps its summed gradient and its parameters."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(out_path, dtype, B, n_steps):
    import dist as D
    import engine as E
    import hashinit
    import synth
    from schema import FULL, state_shapes
    world, rank, _ = D.init("gloo")
    dev = "cuda:0"
    torch.cuda.set_device(0)
    nl, T_, n_news = 2, 2, 3000
    cfg = E.EngineConfig(n_layers=nl, trainable_layers=(0, 1), num_teachers=T_)
    b = B // world
    eng = E.Engine(cfg, dev, max_batch=b, dtype=dtype)
    eng.load_state_dict(hashinit.init_state_dict(7, state_shapes(FULL, nl, cfg.D, T_)))
    D.broadcast_flat([eng.flat[True], eng.flat[False]])
    eng.refresh_shadows(all_layers=True)
    gs = D.GradSync(eng.flat_g, eng.bucket_ranges(), world, force=True)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    comb, tabs = d(synth.news_table(7, n_news, cfg.L)), d(synth.teacher_tables(7, T_, n_news, cfg.D))
    hidx, mask, cidx, label = [d(x) for x in synth.impressions(8, B * n_steps, n_news, cfg.U, cfg.C)]
    grads = []
    for st in range(n_steps):
        s = slice(st * B + rank * b, st * B + (rank + 1) * b)
        eng.forward_indexed(comb, hidx[s], mask[s], cidx[s], label[s], tabs)
        eng.backward(after_bucket=gs.launch if world > 1 else None)
        gs.wait()
        grads.append((eng.flat_g * gs.scale).cpu().numpy().copy())
        eng.step(lr=1e-4, grad_scale=gs.scale)
    torch.cuda.synchronize()
    np.savez(out_path % rank, grads=np.stack(grads), params=eng.flat[True].cpu().numpy(), gscale=np.float32(eng.gscale),
             head0=np.int64(eng.off(E.PFX + "dense.weight")))
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
