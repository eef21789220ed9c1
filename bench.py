#!/usr/bin/env python
"""Headline benchmark: training impressions/sec of the Tiny-NewsRec hot path on N MI355X.

One "step" = Model.forward + backward + AMSGrad (+ gradient all-reduce for N > 1) over one batch of
B=32 synthetic MIND-shaped impressions per GPU (run.py:178-195), 4-layer student + 4-teacher KD.
Inputs (token table, teacher tables, pre-drawn impression indices) are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0 (contract in the task statement).

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

PEAK_BF16 = 2.5e15      # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
N_NEWS = 51282          # MIND-small-train sized news table (+ pad row 0)


def flops_per_impression(n_layers, n_trainable, L=30, H=768, S=55):
    """23.51 GFLOP per layer-forward per impression, backward = 2x on trainable
    layers, + 1.7 GFLOP heads."""
    f_tok = 24 * H * H + 4 * L * H
    return f_tok * S * L * (n_layers + 2 * n_trainable) + 1.7e9


def cpu_baseline(cfg_kw, seed):
    """The oracle (numpy port of the reference step) timed on this host's cores, bounded sample."""
    import hashinit
    import synth
    from oracle import newsrec_oracle as O
    from tests.helpers import FULL, state_shapes
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    nl, tr, T_ = cfg_kw["n_layers"], cfg_kw["trainable_layers"], cfg_kw["num_teachers"]
    B, U, C, L, D = 2, 50, 5, 30, 256
    P = hashinit.init_state_dict(seed, state_shapes(FULL, nl, D, T_))
    comb = synth.news_table(seed, 2000, L).astype(np.int64)
    hidx, mask, cidx, label = synth.impressions(seed, B, 2000, U, C)
    tt = synth.teacher_tables(seed, T_, 2000, D)
    cfg = dict(n_layers=nl, heads=12, trainable_layers=list(tr), user_log_mask=False, temperature=1.0, coef=0.2)
    inp = (comb[hidx], mask, comb[cidx], label, [tt[i][hidx] for i in range(T_)], [tt[i][cidx] for i in range(T_)])
    state = {}

    def step():
        out = O.model_fwd(P, cfg, *inp)
        G = O.model_bwd(P, cfg, out)
        for k, g in G.items():
            st = state.setdefault(k, [np.zeros_like(g), np.zeros_like(g), np.zeros_like(g)])
            O.amsgrad_step(P[k], g, st[0], st[1], st[2], 1, lr=1e-4)

    step()
    n, t0 = 0, time.time()
    while n < 3 or (time.time() - t0 < 10.0 and n < 20):
        step()
        n += 1
    dt = time.time() - t0
    return {"value": round(B * n / dt, 3), "unit": "impressions/s", "cores": int(cores), "kind": "port",
            "sample": "oracle (numpy fp32 port of the reference step: fwd+bwd+AMSGrad), same %d-layer + %d-teacher "
                      "model, B=%d impressions x %d steps" % (nl, T_, B, n)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="impressions per GPU per step (demo.sh:8)")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--trainable", type=int, nargs="+", default=None)
    ap.add_argument("--teachers", type=int, default=4)
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16", help="16-bit activation type (same MFMA rate)")
    ap.add_argument("--force-dp", action="store_true", help="run the RCCL broadcast / bucketed all-reduce path even at world size 1")
    ap.add_argument("--dedup", choices=["off", "also", "only"], default="also",
                    help="in-batch news de-duplication (dedup.py): 'also' times it in a second loop and reports it beside "
                         "the headline (which stays un-deduplicated), 'only' makes it the headline")
    ap.add_argument("--frozen-cache", action="store_true",
                    help="with --dedup only: also take the frozen lower layers from the per-news cache in the headline loop "
                         "(run.py's default mode; not the BASELINE workload, which recomputes every layer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    a = ap.parse_args()

    import dist as D
    import engine as E
    import hashinit
    import synth
    import tnr_hip as T
    from tests.helpers import FULL, state_shapes

    if a.force_dp and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    world, rank, local = D.init()
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world)
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    trainable = tuple(a.trainable) if a.trainable else (a.layers - 2, a.layers - 1)
    cfg_kw = dict(n_layers=a.layers, trainable_layers=trainable, num_teachers=a.teachers)
    cfg = E.EngineConfig(**cfg_kw)
    eng = E.Engine(cfg, dev, max_batch=a.batch, dtype=a.dtype)
    seed = 1234
    eng.load_state_dict(hashinit.init_state_dict(seed, state_shapes(FULL, a.layers, cfg.D, a.teachers)))
    D.broadcast_flat([eng.flat[True], eng.flat[False]], force=a.force_dp)
    eng.refresh_shadows(all_layers=True)

    B, K, W = a.batch, a.steps, a.warmup
    comb = torch.from_numpy(synth.news_table(seed, N_NEWS, cfg.L)).to(dev)
    tables = torch.from_numpy(synth.teacher_tables(seed, max(a.teachers, 1), N_NEWS, cfg.D)).to(dev)
    hidx, mask, cidx, label = [torch.from_numpy(x).to(dev) for x in
                               synth.impressions(seed + 1 + rank, (K + W) * B, N_NEWS, cfg.U, cfg.C)]
    gs = D.GradSync(eng.flat_g, eng.bucket_ranges(), world, force=a.force_dp)

    plans = None
    if a.dedup != "off":
        from dedup import build_plan
        hn, cn = hidx.cpu().numpy(), cidx.cpu().numpy()
        plans = [build_plan(hn[i * B:(i + 1) * B], cn[i * B:(i + 1) * B]) for i in range(K + W)]
        distinct = float(np.mean([p.n_unique / p.n_slots if p is not None else 1.0 for p in plans]))
        encoded = float(np.mean([p.n_enc / p.n_slots if p is not None else 1.0 for p in plans]))
        plans = [p.to(dev) if p is not None else None for p in plans]
    use_plan = [a.dedup == "only"]

    def one_step(i):
        s = slice(i * B, (i + 1) * B)
        eng.forward_indexed(comb, hidx[s], mask[s], cidx[s], label[s], tables if a.teachers else None,
                            plans[i] if use_plan[0] else None)
        eng.backward(after_bucket=gs.launch if (world > 1 or a.force_dp) else None)
        gs.wait()
        eng.step(lr=1e-4, grad_scale=gs.scale)

    def timed_loop():
        for i in range(W):
            one_step(i)
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(W, W + K):
            one_step(i)
        torch.cuda.synchronize()
        D.barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        return float(D.all_reduce_max(t).item())

    def reset():
        eng.load_state_dict(hashinit.init_state_dict(seed, state_shapes(FULL, a.layers, cfg.D, a.teachers)))   # same start
        eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_vmax.zero_(); eng.step_count = 0

    dt_dedup = dt_cache = None
    if a.dedup == "also":
        # extra measurements first (same batches, identical results): each distinct news of a batch encoded once, and on
        # top of that the frozen lower layers taken from a per-news cache; W warm-up + K timed steps each
        use_plan[0] = True
        dt_dedup = timed_loop()
        reset()
        if eng.build_frozen_cache(comb):
            dt_cache = timed_loop()
        use_plan[0] = False
        reset()                                  # also drops the cache: the headline recomputes every layer every step
    if a.dedup == "only" and a.frozen_cache:
        eng.build_frozen_cache(comb)
    # headline: W untimed warm-up steps, then exactly K timed steps
    for i in range(W):
        one_step(i)
    if not a.no_kernel_timing:
        TKEY = "tnr_gemm_nt_ex_f16" if a.dtype == "fp16" else "tnr_gemm_nt_ex"
        T.TIMED[TKEY] = []
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        one_step(i)
    torch.cuda.synchronize()
    D.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dt = float(D.all_reduce_max(dt).item())
    loss = float(eng.total_loss().item())
    rec = T.TIMED.pop("tnr_gemm_nt_ex_f16", None) or T.TIMED.pop("tnr_gemm_nt_ex", None)

    if rank == 0:
        value = world * B * K / dt
        fpi = flops_per_impression(a.layers, len(trainable))
        out = {
            "metric": "training impressions/sec", "value": round(value, 2), "unit": "impressions/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * dt / K, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "Tiny-NewsRec %d-layer student (train %s) + %d-teacher KD (title-emb MSE + soft-CE), "
                                   "fwd+bwd+AMSGrad%s" % (a.layers, list(trainable), a.teachers,
                                                           " + RCCL grad all-reduce" if world > 1 else ""),
                       "batch_per_gpu": B, "global_batch": B * world, "user_log_length": cfg.U, "candidates": cfg.C,
                       "num_words_title": cfg.L, "news_dim": cfg.D, "n_news": N_NEWS,
                       "parallelism": "dp%d" % world, "final_loss": round(loss, 5)},
            "model_flops_per_impression": fpi,
            "mfma_frac_whole_step": round(fpi * value / (world * PEAK_BF16), 4),
        }
        if rec:
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in rec)
            fl = sum(w for _, _, w in rec)
            ach = fl / (ms * 1e-3) / 1e12
            traffic = None
            pj = os.path.join(ROOT, "profiles", "r01_gemm_nt_pmc.json")
            if os.path.exists(pj):
                traffic = json.load(open(pj)).get("hbm_bytes_per_launch")
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt256x256_kernel (bf16 MFMA, every forward/dgrad Linear GEMM, incl. fused epilogues)",
                               "achieved": round(ach, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                               "frac": round(ach * 1e12 / PEAK_BF16, 4), "traffic": traffic,
                               "launches": len(rec), "avg_launch_us": round(1e3 * ms / len(rec), 2),
                               "algorithmic_flops_per_launch": fl / len(rec)}
        else:
            out["roofline"] = None
        if a.dedup != "off":
            out["dedup"] = {"in_headline": a.dedup == "only", "frozen_layer_cache_in_headline": bool(a.dedup == "only" and a.frozen_cache), "distinct_news_frac": round(distinct, 4),
                            "encoded_frac": round(encoded, 4),
                            "note": "identical outputs; FLOPs per impression above stay un-deduplicated (SURVEY 8-d)"}
            if dt_dedup is not None:
                out["dedup"].update(value=round(world * B * K / dt_dedup, 2), ms_per_step=round(1e3 * dt_dedup / K, 4))
            if dt_cache is not None:
                out["dedup"]["with_frozen_layer_cache"] = {"value": round(world * B * K / dt_cache, 2),
                                                           "ms_per_step": round(1e3 * dt_cache / K, 4)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg_kw, seed)
        print(json.dumps(out), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
