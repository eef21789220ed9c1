"""One data-parallel worker of tests/test_dp_gpu.py (started once per rank, all on cuda:0, torch.distributed over gloo so
that two ranks can share one GPU; device buckets are staged through host memory by dist.GradSync).

Each rank takes impressions [rank * B/W, (rank+1) * B/W) of a B-impression batch, runs Engine.forward / backward with the
bucketed all-reduce launched from the backward's after_bucket hook exactly as run.py / bench.py do, steps AMSGrad with the
1/W scale THROUGH Engine.step(sync=...) -- bucket by bucket, each optimiser slice behind its own in-flight collective
(GradSync keeps the gloo all-reduces asynchronous on pinned host mirrors) -- and dumps its summed gradient and its parameters.
mode "stage1": the same for post_train_kd.py's step (Stage1Engine: title pass writes the gradients, body pass accumulates and
fires the buckets)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_stage1(out_path, dtype, B, n_steps):
    """post_train_kd.py's data-parallel step on a small Stage1Engine: 2 layers, 2 teachers, 1 + 3 titles of 24 tokens, bodies
    of 64; plain two-rate Adam as the notebook's cell 18."""
    import dist as D
    import hashinit
    import synth
    from schema import FULL, state_shapes
    from stage1 import Stage1Engine
    world, rank, _ = D.init("gloo")
    dev = "cuda:0"
    torch.cuda.set_device(0)
    nl, T_, n_docs, K, Lt, Lb = 2, 2, 500, 3, 24, 64
    b = B // world
    eng = Stage1Engine(n_layers=nl, trainable_layers=(0, 1), num_teachers=T_, npratio=K, title_len=Lt, body_len=Lb, device=dev,
                       batch=b, dtype=dtype)
    eng.load_state_dict({k: hashinit.init_tensor(11, k, tuple(shp)) for k, shp in eng.shapes.items()})
    t_eng = eng.title
    D.broadcast_flat([t_eng.flat[True], t_eng.flat[False]])
    t_eng.refresh_shadows(all_layers=True)
    eng.body.refresh_rel()
    gs = D.GradSync(t_eng.flat_g, eng.bucket_ranges(), world, force=True)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    titles, bodies = d(synth.news_table(11, n_docs, Lt)), d(synth.news_table(12, n_docs, Lb))
    tt, tb = d(synth.teacher_tables(11, T_, n_docs, 256)), d(synth.teacher_tables(12, T_, n_docs, 256))
    rs = np.random.RandomState(13)
    idx = d(rs.randint(1, n_docs, (B * n_steps, K + 1)).astype(np.int32))
    label = torch.zeros(b, dtype=torch.int64, device=dev)
    grads = []
    for st in range(n_steps):
        s = slice(st * B + rank * b, st * B + (rank + 1) * b)
        eng.forward_indexed(titles, bodies, idx[s], label, tt, tb)
        eng.backward(after_bucket=gs.launch if world > 1 else None)
        if st == 0:
            gs.wait()
            grads.append((t_eng.flat_g * gs.scale).cpu().numpy().copy())
            eng.step(1e-5, grad_scale=gs.scale, lr_bert=1e-6, amsgrad=False)
        else:
            eng.step(1e-5, grad_scale=gs.scale, lr_bert=1e-6, amsgrad=False, sync=gs)
            assert not gs.pending
            grads.append((t_eng.flat_g * gs.scale).cpu().numpy().copy())
    torch.cuda.synchronize()
    import engine as E
    np.savez(out_path % rank, grads=np.stack(grads), params=t_eng.flat[True].cpu().numpy(), gscale=np.float32(t_eng.gscale),
             head0=np.int64(t_eng.off(E.PFX + "dense.weight")))
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


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
        if world > 1:
            assert sorted(gs.pending) == list(range(len(gs.ranges)))      # every bucket's collective is in flight
        if st == 0:
            # what the buckets sum to, read BEFORE the optimiser (wait here, then step without sync); later steps take the
            # bucket-by-bucket path of run.py / bench.py
            gs.wait()
            grads.append((eng.flat_g * gs.scale).cpu().numpy().copy())
            eng.step(lr=1e-4, grad_scale=gs.scale)
        else:
            eng.step(lr=1e-4, grad_scale=gs.scale, sync=gs)
            assert not gs.pending
            grads.append((eng.flat_g * gs.scale).cpu().numpy().copy())
    torch.cuda.synchronize()
    np.savez(out_path % rank, grads=np.stack(grads), params=eng.flat[True].cpu().numpy(), gscale=np.float32(eng.gscale),
             head0=np.int64(eng.off(E.PFX + "dense.weight")))
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    (run_stage1 if len(sys.argv) > 5 and sys.argv[5] == "stage1" else run)(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
