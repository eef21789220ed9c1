#!/usr/bin/env python
"""Headline benchmark: training impressions/sec of the Tiny-NewsRec hot path on N MI355X.

One "step" = Model.forward + backward + AMSGrad (+ gradient all-reduce for N > 1) over one batch of
B=32 synthetic MIND-shaped impressions per GPU (run.py:178-195), 4-layer student + 4-teacher KD.
Inputs (token table, teacher tables, pre-drawn impression indices) are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0 (contract in the task statement).

  python bench.py --gpus 1 --steps 200 --warmup 20      (the defaults: SURVEY 8-d)
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
CAL_REF_US = 232.0      # `box` reference = the MFMA-heavy calibration launch in step context, median over the round-6 build leases (README.md)
EVENT_EVERY = 4         # roofline leg: every 4th timed step has its NT GEMM launches bracketed by HIP events


def flops_per_impression(n_layers, n_trainable, L=30, H=768, S=55):
    """23.51 GFLOP per layer-forward per impression, backward = 2x on trainable
    layers, + 1.7 GFLOP heads."""
    f_tok = 24 * H * H + 4 * L * H
    return f_tok * S * L * (n_layers + 2 * n_trainable) + 1.7e9


def host_cpu():
    """(model name, physical cores) from lscpu; falls back to os.cpu_count()."""
    import subprocess
    model, cores = "unknown", os.cpu_count() or 1
    try:
        info = {}
        for ln in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if ":" in ln:
                k, v = ln.split(":", 1)
                info[k.strip()] = v.strip()
        model = info.get("Model name", model)
        cores = int(info["Core(s) per socket"]) * int(info["Socket(s)"])
    except Exception:
        pass
    return model, max(1, cores)


def cpu_baseline(cfg_kw, seed):
    """The reference step on this host's CPU cores, bounded sample: oracle/torch_port.py -- a torch-CPU
    port of the reference loop body (forward, autograd backward, Adam(amsgrad)), pinned to the reference's goldens in
    tests/test_oracle_golden.py -- because /root/reference itself cannot travel to the GPU box.  The thread count is SWEPT
    (physical cores, then 128, 64, 32, 16, 8 below that; one warm-up + one timed step each at B/2) and the best one is timed at B: the
    title-sized GEMMs (L = 30) do not scale to 128 threads, and a baseline should be the best the host can do.  `value` is the
    headline model at B=16; BASELINE configs[0] (PLM-NR 2-layer, B=16, fp32: the reference's own CPU-runnable case) beside it."""
    import hashinit
    import synth
    from oracle import torch_port as TP
    from schema import FULL, state_shapes
    model, cores = host_cpu()
    U, C, L, D = 50, 5, 30, 256
    cand = sorted({t for t in (8, 16, 32, 64, 128, 256) if t < cores} | {cores})

    def run(nl, tr, T_, B, budget_s):
        P = hashinit.init_state_dict(seed, state_shapes(FULL, nl, D, T_))
        comb = synth.news_table(seed, 2000, L).astype(np.int64)
        tt = synth.teacher_tables(seed, max(T_, 1), 2000, D)
        cfg = dict(n_layers=nl, heads=12, trainable_layers=list(tr), user_log_mask=False, temperature=1.0, coef=0.2 if T_ else 1.0)
        trn = TP.Trainer(P, cfg, lr=1e-4)

        def batch(b):
            hidx, mask, cidx, label = synth.impressions(seed, b, 2000, U, C)
            return (comb[hidx], mask, comb[cidx], label, [tt[i][hidx] for i in range(T_)], [tt[i][cidx] for i in range(T_)])

        t_start = time.time()
        half, sweep = batch(B // 2), {}
        for th in sorted(cand, reverse=True):             # sweep at B/2 (half the cost per sample): one warm-up + one timed step
            if len(sweep) >= 2 and time.time() - t_start > 0.55 * budget_s:
                break
            torch.set_num_threads(th)
            trn.step(*half)
            t0 = time.time()
            trn.step(*half)
            sweep[th] = (B // 2) / (time.time() - t0)
        best = max(sweep, key=sweep.get)
        torch.set_num_threads(best)
        full = batch(B)
        trn.step(*full)                                   # SURVEY 8-d / BASELINE.md section 3: 2 warm-up + 5 timed steps at the
        trn.step(*full)                                   # reported batch size (fewer timed ones only if a slow host runs out of budget)
        n, t0 = 0, time.time()
        while n < 5 and (n < 1 or time.time() - t_start < budget_s):
            trn.step(*full)
            n += 1
        return B * n / (time.time() - t0), best, n, {str(k): round(x, 3) for k, x in sorted(sweep.items())}

    nl, tr, T_ = cfg_kw["n_layers"], cfg_kw["trainable_layers"], cfg_kw["num_teachers"]
    v_head, th_head, n_head, sw_head = run(nl, tr, T_, 16, 100.0)
    v_c0, th_c0, n_c0, sw_c0 = run(2, (0, 1), 0, 16, 70.0)
    return {"value": round(v_head, 3), "unit": "impressions/s", "cores": int(th_head), "host_cores": int(cores), "cpu_model": model,
            "kind": "port", "threads_sweep": sw_head,
            "sample": "oracle/torch_port.py (torch-CPU port of the reference step: fwd + autograd bwd + Adam(amsgrad), fp32), same "
                      "%d-layer + %d-teacher model, B=16 impressions; threads swept over %s (1 warm-up + 1 timed step each at B=8), best = %d "
                      "threads: 2 warm-up + %d timed steps at B=16 (SURVEY 8-d protocol)" % (nl, T_, list(sw_head), th_head, n_head),
            "configs0_plmnr_2layer_b16": {"value": round(v_c0, 3), "unit": "impressions/s", "cores": int(th_c0), "threads_sweep": sw_c0,
                                         "sample": "BASELINE configs[0]: PLM-NR 2-layer (train 0,1), B=16, fp32; same sweep, best = %d "
                                                   "threads, 2 warm-up + %d timed steps" % (th_c0, n_c0)}}


FAMILIES = (("nt_gemm", ("tnr_gemm_nt",)), ("weight_gradient", ("tnr_gemm_tn_wgrad",)), ("attention", ("tnr_attn_",)),
            ("layernorm_embed", ("tnr_ln_", "tnr_embed_")),
            ("optimiser", ("tnr_amsgrad", "tnr_adam", "tnr_refresh", "tnr_grad_nonfinite", "tnr_cast_")))


def family_of(name):
    for fam, prefixes in FAMILIES:
        if name.startswith(prefixes):
            return fam
    return "heads_kd_misc"


class BoxProbe:
    """What this BOX does with fixed pieces of work, so that two bench lines can be told apart into box and build: two fixed
    persistent NT launches on random operands, plain epilogue, M = 52 800 - "store_heavy" N = K = 768 (62 GFLOP, 81 MB out: the shape
    the round-5 review named) and "mfma_heavy" N = 768, K = 3072 (249 GFLOP behind the same output) - and the shader clock the chip
    holds under each (tnr_gemm_clock_stamps: cycles / 100 MHz ticks per workgroup).  launch() is called right BEHIND a training
    step (the step breakdown's extra steps, after the timed region): the chip is then in the power state the step's own GEMMs see -
    the same launch repeated back to back for milliseconds runs at another clock (1.5 against 2.06 GHz in-step on one of this
    round's boxes).  Over the round-6 leases the store-heavy launch did NOT order the headlines (77.8 ... 80.7 us, the slowest on
    the fastest box); the MFMA-heavy one is what `headline_normalised` uses."""
    SHAPES = (("store_heavy", 768, 768), ("mfma_heavy", 768, 3072))

    def __init__(self, T, dev, dtype):
        self.T, self.sfx = T, "_f16" if dtype == "fp16" else ""
        td = torch.float16 if dtype == "fp16" else torch.bfloat16
        self.M = 52800
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        self.ops = {}
        for name, N, K in self.SHAPES:
            self.ops[name] = ((torch.randn((self.M, K), device=dev, generator=g) * 0.5).to(td),
                              (torch.randn((N, K), device=dev, generator=g) * 0.05).to(td), torch.zeros((self.M, N), device=dev, dtype=td))
        self.ev = {name: [] for name, _, _ in self.SHAPES}
        self.stamps = {name: [torch.zeros((256, 2), device=dev, dtype=torch.int64) for _ in range(16)] for name, _, _ in self.SHAPES}

    def launch(self):
        T, M = self.T, self.M
        for name, N, K in self.SHAPES:
            a_, b_, c_ = self.ops[name]
            i = len(self.ev[name])
            T.lib().tnr_gemm_clock_stamps(self.stamps[name][i].data_ptr(), 256)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            T.call("tnr_gemm_nt_ex" + self.sfx, a_, K, b_, K, c_, N, M, N, K, None, None, 0, None, 0, 0, None)
            e1.record()
            self.ev[name].append((e0, e1))
        T.lib().tnr_gemm_clock_stamps(None, 0)

    def result(self):
        T, M = self.T, self.M
        out = {"protocol": "each launch once right behind each of the step breakdown's extra training steps; medians"}
        for name, N, K in self.SHAPES:
            us = sorted(1e3 * e0.elapsed_time(e1) for e0, e1 in self.ev[name])
            med = us[len(us) // 2]
            mhz = []
            for st in self.stamps[name][:len(us)]:
                st = st.cpu().numpy()
                ok = st[:, 1] > 0
                if ok.any():
                    mhz.append(float(np.median(100.0 * st[ok, 0] / st[ok, 1])))
            out[name] = {"M": M, "N": N, "K": K, "epilogue": "plain", "samples": len(us), "us": round(med, 2),
                         "us_min_max": [round(us[0], 2), round(us[-1], 2)], "tflops": round(2.0 * M * N * K / med / 1e6, 1),
                         "route": T.query("tnr_gemm_nt_route" + self.sfx, M, N, K, 0),
                         "mfma_clock_mhz": round(float(np.median(mhz)), 0) if mhz else None}
        return {"calibration_launches": out}


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start N workers (one per GPU) under torch.distributed.run as a CHILD
    process and pass its exit code on.  Nothing in this process has touched the GPU yet (no HIP call, no
    torch.cuda.is_available()), and nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)




def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # 20 warm-up + 200 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="impressions per GPU per step (demo.sh:8)")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--trainable", type=int, nargs="+", default=None)
    ap.add_argument("--teachers", type=int, default=4)
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="fp16",
                    help="16-bit activation type of the headline (same MFMA rate; fp16 meets the north-star 1e-3 logit / loss "
                         "tolerance, bf16 does not -- DESIGN.md section 2)")
    ap.add_argument("--no-other-dtype", action="store_true", help="skip the extra timed loop in the other 16-bit type")
    ap.add_argument("--force-dp", action="store_true", help="run the RCCL broadcast / bucketed all-reduce path even at world size 1")
    ap.add_argument("--wgrad-units", type=int, default=0,
                    help="work units per workgroup of the weight-gradient kernel; 0 = 1 on one GPU, 2 under data parallelism (a "
                         "workgroup kept off its CU by an overlapped collective then costs half a round, DESIGN.md: multi-GPU)")
    ap.add_argument("--dp-buckets", type=int, choices=[5, 3], default=None,
                    help="gradient collectives per step: 5 = heads + FFN / attention block of each trainable layer, 3 = heads + one per layer "
                         "(default: 5 on one GPU; with more than one rank chosen by the warm-up autotune, see --no-dp-autotune)")
    ap.add_argument("--no-dp-autotune", action="store_true",
                    help="more than one rank: skip the pre-pass that times {wgrad units 2, 1} x {5, 3 buckets} for 3 + 10 untimed steps each "
                         "and runs the headline with the fastest (max over ranks; knobs given explicitly are kept) - defaults 2 / 5 then")
    ap.add_argument("--dp-algo", choices=["allreduce", "rs_ag", "native", "native_rs_ag"], default="allreduce",
                    help="one all-reduce per bucket, or reduce-scatter + all-gather (direct exchange on the xGMI mesh); native*: the same "
                         "two through the library's own C ABI (tnr_comm_*: RCCL bound inside libtnr_hip.so) instead of ProcessGroupNCCL")
    ap.add_argument("--dp-sweep", action="store_true",
                    help="data-parallel runs only: after the headline also time {wgrad units 1, 2} x {5, 3 buckets} x {allreduce, rs_ag} "
                         "(30 steps each) and report them in the dp object")
    ap.add_argument("--dedup", choices=["off", "also", "only"], default="also",
                    help="in-batch news de-duplication (dedup.py): 'also' times it in a second loop and reports it beside "
                         "the headline (which stays un-deduplicated), 'only' makes it the headline")
    ap.add_argument("--frozen-cache", action="store_true",
                    help="with --dedup only: also take the frozen lower layers from the per-news cache in the headline loop "
                         "(run.py's default mode; not the BASELINE workload, which recomputes every layer)")
    ap.add_argument("--gemm-opt", action="append", default=[], metavar="KEY=INT",
                    help="tools only: tnr_gemm_set_option(KEY, INT) before the run (A/B profiles; the default is the shipped configuration)")
    ap.add_argument("--no-larger-batch", action="store_true", help="skip the extra timed loop at 4 x the per-GPU batch")
    ap.add_argument("--no-configs", action="store_true", help="skip the timed legs of the other BASELINE.json configurations")
    ap.add_argument("--leg", default=None, help="run ONE of the other configurations alone (--warmup / --steps apply) and print its object: "
                                                "'configs[1]', 'configs[2]', 'configs[4]', 'configs[4] stage 1', 'stage 1 notebook shape'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--concat-batches", action="store_true",
                    help="tests only: the ranks' batches of a step are consecutive slices of ONE batch of world x B impressions drawn from "
                         "a common seed (so that a one-rank run with --batch world x B sees the concatenation); default: per-rank seeds")
    a = ap.parse_args()
    # tests only (tests/test_dropin_gpu.py): TNR_BENCH_BACKEND=gloo runs the world > 1 path - self_launch, per-rank batches, bucketed
    # all-reduce with its waits, max over ranks, the dp object, --dp-sweep - as N processes over gloo, TNR_BENCH_SHARE_GPU=1 puts them
    # all on GPU 0, TNR_BENCH_DUMP_PARAMS=<file.npy> saves a sample of the trained parameters.  No throughput claim comes from it.
    backend = os.environ.get("TNR_BENCH_BACKEND") or None
    share_gpu = os.environ.get("TNR_BENCH_SHARE_GPU") == "1"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))

    import dist as D
    import engine as E
    import hashinit
    import synth
    import tnr_hip as T
    from schema import FULL, state_shapes

    for kv in a.gemm_opt:
        k_, v_ = kv.split("=")
        assert T.lib().tnr_gemm_set_option(k_.encode(), int(v_)) == 0, kv
    if a.force_dp and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    world, rank, local = D.init(backend)
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (a.gpus, world))
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    trainable = tuple(a.trainable) if a.trainable else (a.layers - 2, a.layers - 1)
    cfg_kw = dict(n_layers=a.layers, trainable_layers=trainable, num_teachers=a.teachers)
    cfg = E.EngineConfig(**cfg_kw)
    seed = 1234
    B, K, W = a.batch, a.steps, a.warmup
    use_dp = world > 1 or a.force_dp
    wgrad_units = a.wgrad_units or (2 if use_dp else 1)
    tune_units, tune_buckets = a.wgrad_units == 0, a.dp_buckets is None
    if a.dp_buckets is None:
        a.dp_buckets = 5
    E.Engine.WGRAD_UNITS = wgrad_units           # before any engine is built: the slab workspace is sized by it
    comb = torch.from_numpy(synth.news_table(seed, N_NEWS, cfg.L)).to(dev)
    tables = torch.from_numpy(synth.teacher_tables(seed, max(a.teachers, 1), N_NEWS, cfg.D)).to(dev)
    if a.concat_batches:
        imp = [x.reshape((K + W, world, B) + x.shape[1:])[:, rank].reshape(((K + W) * B,) + x.shape[1:])
               for x in synth.impressions(seed + 1, (K + W) * B * world, N_NEWS, cfg.U, cfg.C)]
    else:
        imp = synth.impressions(seed + 1 + rank, (K + W) * B, N_NEWS, cfg.U, cfg.C)
    hidx, mask, cidx, label = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in imp]
    plans = None
    if a.dedup != "off":
        from dedup import build_plan
        hn, cn = hidx.cpu().numpy(), cidx.cpu().numpy()
        plans = [build_plan(hn[i * B:(i + 1) * B], cn[i * B:(i + 1) * B]) for i in range(K + W)]
        distinct = float(np.mean([p.n_unique / p.n_slots if p is not None else 1.0 for p in plans]))
        encoded = float(np.mean([p.n_enc / p.n_slots if p is not None else 1.0 for p in plans]))
        plans = [p.to(dev) if p is not None else None for p in plans]
    init_sd = hashinit.init_state_dict(seed, state_shapes(FULL, a.layers, cfg.D, a.teachers))

    def build(dtype):
        eng = E.Engine(cfg, dev, max_batch=B, dtype=dtype)
        eng.load_state_dict(init_sd)
        D.broadcast_flat([eng.flat[True], eng.flat[False]], force=a.force_dp)
        eng.refresh_shadows(all_layers=True)
        return eng, D.GradSync(eng.flat_g, eng.bucket_ranges(), world, force=a.force_dp, timing=use_dp,
                               merge_layers=a.dp_buckets == 3, algo=a.dp_algo)

    def reset(eng):
        eng.load_state_dict(init_sd)                 # same start; also drops the frozen-layer cache
        eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_vmax.zero_(); eng.step_count = 0

    def one_step(eng, gs, i, use_plan):
        s = slice(i * B, (i + 1) * B)
        eng.forward_indexed(comb, hidx[s], mask[s], cidx[s], label[s], tables if a.teachers else None,
                            plans[i] if use_plan else None)
        eng.backward(after_bucket=gs.launch if use_dp else None)
        eng.step(lr=1e-4, grad_scale=gs.scale, sync=gs)      # per bucket: wait for its all-reduce, then its AMSGrad slice

    # The other BASELINE.json configurations, each on its own shapes (never `value`): configs[1] PLM-NR 12-layer fine-tune (train
    # 10-11, bf16; PLM-NR/demo.sh:3-22), configs[2] 4-layer student + ONE teacher, configs[4] 2-layer student + 4 teachers in fp16
    # (Tiny-NewsRec/demo.sh:3-36) and its stage-1 form, titles of 30 / bodies of 128 tokens, 1 + 4 titles per body
    # (Post-train_KD.ipynb cell 4 with BASELINE's lengths), and the notebook's own stage-1 shape: titles of 24 / bodies of 512 tokens,
    # 1 + 9 titles per body (Post-train_KD.ipynb cells 4, 8).  B = 32 per GPU, 3 warm-up + 20 timed steps each, one GPU.
    # `--leg NAME` runs one of them alone (tools/make_profiles.sh: a rocprofv3 trace per configuration).
    legs = {}
    # what each leg's parity test allows and what it measured (tools/parity_measured.py runs the tests and writes the file; the
    # allowances beyond 1e-3 are measured ones)
    PARITY = {}
    import glob as _glob
    for pj in sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_parity_measured.json")), reverse=True)[:1]:
        PARITY = json.load(open(pj))

    def stage2_leg(key, nl, tr, T_, dtype, what, W2=3, K2=20):
        try:
            c2 = E.EngineConfig(n_layers=nl, trainable_layers=tr, num_teachers=T_)
            e2 = E.Engine(c2, dev, max_batch=B, dtype=dtype)
            e2.load_state_dict(hashinit.init_state_dict(seed, state_shapes(FULL, nl, c2.D, T_)))
            e2.refresh_shadows(all_layers=True)

            def st(i):
                s2 = slice((i % (K + W)) * B, (i % (K + W) + 1) * B)
                e2.forward_indexed(comb, hidx[s2], mask[s2], cidx[s2], label[s2], tables[:T_] if T_ else None)
                e2.backward()
                e2.step(lr=1e-4)
            for i in range(W2):
                st(i)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(W2, W2 + K2):
                st(i)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t2
            f2 = flops_per_impression(nl, len(tr))
            legs[key] = {"workload": what, "dtype": dtype, "batch_per_gpu": B, "steps": K2, "value": round(B * K2 / d2, 2),
                         "unit": "impressions/s", "ms_per_step": round(1e3 * d2 / K2, 4), "model_flops_per_impression": f2,
                         "mfma_frac_whole_step": round(f2 * B * K2 / d2 / PEAK_BF16, 4), "final_loss": round(float(e2.total_loss().item()), 5),
                         "parity": PARITY.get(key)}
            del e2
        except Exception as ex:                  # an informational leg must never cost the headline line
            legs[key] = {"workload": what, "error": repr(ex)[:300]}
        torch.cuda.empty_cache()

    def stage1_leg(key, Lt, Lb, Kn, what, W2=3, K2=20):
        try:
            from stage1 import Stage1Engine
            nd = 20000
            s1 = Stage1Engine(n_layers=2, trainable_layers=(0, 1), num_teachers=4, npratio=Kn, title_len=Lt, body_len=Lb, device=dev,
                              batch=B, dtype="fp16")
            for knob in ("chain_wgrad", "two_streams", "joint", "joint_streams", "joint_group_wgrad"):              # A/B switches of tools/ only (defaults = the class's)
                if os.environ.get("TNR_S1_" + knob.upper()) is not None:
                    setattr(s1, knob, os.environ["TNR_S1_" + knob.upper()] == "1")
            s1.load_state_dict({k: torch.from_numpy(hashinit.init_tensor(seed, k, tuple(sh))) for k, sh in s1.shapes.items()})
            s1.title.refresh_shadows(all_layers=True)
            s1.body.refresh_rel()
            d_title = torch.from_numpy(synth.news_table(11, nd - 1, Lt)).to(dev)
            d_body = torch.from_numpy(synth.news_table(12, nd - 1, Lb, mean_len=0.6 * Lb, std_len=0.25 * Lb)).to(dev)
            d_tt = torch.from_numpy(np.ascontiguousarray(synth.teacher_tables(13, 4, nd - 1, s1.cfg_t.D))).to(dev)
            d_tb = torch.from_numpy(np.ascontiguousarray(synth.teacher_tables(14, 4, nd - 1, s1.cfg_t.D))).to(dev)
            rs = np.random.RandomState(seed)
            pidx = torch.from_numpy(rs.randint(1, nd, ((W2 + K2) * B, 1 + Kn)).astype(np.int32)).to(dev)
            lab1 = torch.zeros(B, dtype=torch.int64, device=dev)
            pcol = pidx[:, 0].contiguous()                 # the positives' (= bodies') indices, as the loader hands them over

            def st1(i):
                s1.forward_indexed(d_title, d_body, pidx[i * B:(i + 1) * B], lab1, d_tt, d_tb, body_idx=pcol[i * B:(i + 1) * B])
                s1.backward()
                s1.step(1e-5, lr_bert=1e-6, amsgrad=False)
            for i in range(W2):
                st1(i)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(W2, W2 + K2):
                st1(i)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t2
            H_ = 768
            ftok = lambda L_: 24 * H_ * H_ + 4 * L_ * H_
            f1 = (ftok(Lt) * (1 + Kn) * Lt + ftok(Lb) * Lb) * (2 + 2 * 2) + 0.2e9      # per (body, 1 + Kn titles) pair group, 2 layers both trainable
            legs[key] = {"workload": what, "dtype": "fp16", "batch_per_gpu": B, "steps": K2, "value": round(B * K2 / d2, 2), "unit": "pairs/s",
                         "ms_per_step": round(1e3 * d2 / K2, 4), "model_flops_per_pair": f1,
                         "mfma_frac_whole_step": round(f1 * B * K2 / d2 / PEAK_BF16, 4),
                         "final_loss": round(float(s1.total_loss().item()), 5), "parity": PARITY.get(key)}
            del s1
        except Exception as ex:
            legs[key] = {"workload": what, "error": repr(ex)[:300]}
        torch.cuda.empty_cache()

    LEGS = {
        "configs[1]": lambda **kw: stage2_leg("configs[1]", 12, (10, 11), 0, "bf16", "PLM-NR 12-layer UniLM teacher fine-tune (train [10, 11]), CE, two-rate AMSGrad path", **kw),
        "configs[2]": lambda **kw: stage2_leg("configs[2]", 4, (2, 3), 1, a.dtype, "Tiny-NewsRec 4-layer student (train [2, 3]) + 1-teacher KD", **kw),
        "configs[4]": lambda **kw: stage2_leg("configs[4]", 2, (0, 1), 4, "fp16", "Tiny-NewsRec 2-layer student (train [0, 1]) + 4-teacher KD", **kw),
        "configs[4] stage 1": lambda **kw: stage1_leg("configs[4] stage 1", 30, 128, 4, "Post-train_KD stage 1: 2-layer student + 4 teachers, 1 + 4 titles of 30 / body of 128 tokens, plain two-rate Adam", **kw),
        "stage 1 notebook shape": lambda **kw: stage1_leg("stage 1 notebook shape", 24, 512, 9, "Post-train_KD stage 1 at the notebook's own shape (cells 4, 8): 2-layer student + 4 teachers, 1 + 9 titles of 24 / body of 512 tokens, plain two-rate Adam", **kw),
    }
    if a.leg:
        LEGS[a.leg](W2=W, K2=K)
        if rank == 0:
            print(json.dumps({"leg": a.leg, **legs[a.leg]}), flush=True)
        return

    timed_rec = []
    clock_stamps = torch.zeros((256, 2), device=dev, dtype=torch.int64)

    def timed_loop(eng, gs, use_plan, time_kernels=None):
        """W untimed warm-up steps, then exactly K timed steps between barrier + synchronize; max over ranks (seconds)."""
        for i in range(W):
            one_step(eng, gs, i, use_plan)
        rec = [] if time_kernels else None
        D.barrier()
        torch.cuda.synchronize()
        gs._events = []                              # exposed all-reduce time: timed steps only
        t0 = time.perf_counter()
        for i in range(W, W + K):
            # the NT GEMM launches of every 4th timed step are bracketed by HIP events on their stream (two event records per
            # launch cost ~1.4 % of the step when every step carries them; sampled, the timed region stays what it measures);
            # the same steps' persistent NT launches stamp their shader clock (two scalar loads per workgroup: `box`)
            if rec is not None and (i - W) % EVENT_EVERY == 0:
                T.TIMED[time_kernels] = rec
                T.lib().tnr_gemm_clock_stamps(clock_stamps.data_ptr(), 256)
            else:
                T.TIMED.pop(time_kernels, None)
                T.lib().tnr_gemm_clock_stamps(None, 0)
            one_step(eng, gs, i, use_plan)
        T.TIMED.pop(time_kernels, None)
        T.lib().tnr_gemm_clock_stamps(None, 0)
        if rec is not None:
            timed_rec.extend(rec)
        torch.cuda.synchronize()
        D.barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        return float(D.all_reduce_max(t).item())

    autotune = None
    if world > 1 and not a.no_dp_autotune and (tune_units or tune_buckets):
        # The two knobs whose best setting depends on what the collectives do to the compute kernels on THIS node (how many CUs
        # RCCL's kernels hold and for how long) cannot be set from a one-GPU lease: every rank times the candidates for 3 + 10
        # untimed steps (the same collective sequence on every rank; max over ranks, so all ranks pick the same) and the headline's
        # W warm-up + K timed steps then run with the fastest.  horovod's own HOROVOD_AUTOTUNE does the like for its fusion buffer.
        K_, W_, cands = K, W, []
        for units in ((2, 1) if tune_units else (wgrad_units,)):
            for nb in ((5, 3) if tune_buckets else (a.dp_buckets,)):
                E.Engine.WGRAD_UNITS, a.dp_buckets = units, nb
                e_, g_ = build(a.dtype)
                K, W = min(10, K_), min(3, W_)
                d_ = timed_loop(e_, g_, False)
                cands.append({"wgrad_units": units, "buckets": nb, "ms_per_step": round(1e3 * d_ / K, 4)})
                del e_, g_
                torch.cuda.empty_cache()
        best = min(cands, key=lambda c: c["ms_per_step"])
        # ... and, with that pair, the persistent GEMM grids sized for 248 / 240 of the CUs: when RCCL's kernels wait for a CU
        # behind a persistent launch that holds every CU's whole register file, leaving them a few CUs can be the cheaper step
        if not any(kv.startswith("cus=") for kv in a.gemm_opt):
            for n_cu in (248, 240):
                E.Engine.WGRAD_UNITS, a.dp_buckets = best["wgrad_units"], best["buckets"]
                T.lib().tnr_gemm_set_option(b"cus", n_cu)
                e_, g_ = build(a.dtype)
                d_ = timed_loop(e_, g_, False)
                cands.append({"wgrad_units": best["wgrad_units"], "buckets": best["buckets"], "cus": n_cu, "ms_per_step": round(1e3 * d_ / K, 4)})
                del e_, g_
                torch.cuda.empty_cache()
            best = min(cands, key=lambda c: c["ms_per_step"])
            T.lib().tnr_gemm_set_option(b"cus", best.get("cus", 0))
        K, W = K_, W_
        wgrad_units, a.dp_buckets = best["wgrad_units"], best["buckets"]
        E.Engine.WGRAD_UNITS = wgrad_units
        autotune = {"candidates": cands, "chosen": best, "note": "3 + 10 untimed steps per candidate before the headline's warm-up; max over ranks; "
                                                                 "cus = CUs the persistent GEMM grids are sized for (absent: all)"}
    eng, gs = build(a.dtype)
    dt_dedup = dt_cache = None
    if a.dedup == "also":
        # extra measurements first (same batches, identical results): each distinct news of a batch encoded once, and on
        # top of that the frozen lower layers taken from a per-news cache; W warm-up + K timed steps each
        dt_dedup = timed_loop(eng, gs, True)
        reset(eng)
        if eng.build_frozen_cache(comb):
            dt_cache = timed_loop(eng, gs, True)
        reset(eng)                               # the headline recomputes every layer every step
    if a.dedup == "only" and a.frozen_cache:
        eng.build_frozen_cache(comb)
    # headline: W untimed warm-up steps, then exactly K timed steps, every NT GEMM launch bracketed by HIP events on its stream
    TKEY = "tnr_gemm_nt_ex_f16" if a.dtype == "fp16" else "tnr_gemm_nt_ex"
    dt = timed_loop(eng, gs, a.dedup == "only", None if a.no_kernel_timing else TKEY)
    rec = timed_rec or None
    box = breakdown = None
    if not a.no_kernel_timing:
        # `box`: the clock the chip held under the LAST stamped NT launch of the timed region + a fixed calibration launch, both
        # OUTSIDE the timed region's clock; `step_breakdown_ms`: 8 more steps (untimed) with EVERY library call bracketed by events
        st_ = clock_stamps.cpu().numpy()
        ok_ = st_[:, 1] > 0
        probe = BoxProbe(T, dev, a.dtype)
        n_bd = min(8, K)
        T.TIMED_ALL = []
        torch.cuda.synchronize()
        tb = 0.0
        for i in range(W, W + n_bd):
            tb0 = time.perf_counter()
            one_step(eng, gs, i, a.dedup == "only")
            all_, T.TIMED_ALL = T.TIMED_ALL, None
            probe.launch()                                   # the calibration launch, in the step's own power state
            T.TIMED_ALL = all_
            torch.cuda.synchronize()
            tb += 1e3 * (time.perf_counter() - tb0) / n_bd
        box = probe.result()
        box["mfma_clock_mhz_under_load"] = round(float(np.median(100.0 * st_[ok_, 0] / st_[ok_, 1])), 0) if ok_.any() else None
        box["note"] = ("under_load: median over the workgroups of the last event-sampled step's last persistent NT launch INSIDE the timed "
                       "region (shader cycles / 100 MHz ticks of a workgroup's life)")
        rec_all, T.TIMED_ALL = T.TIMED_ALL, None
        fam_ms, fam_calls = {}, {}
        for e0_, e1_, nm_ in rec_all:
            f_ = family_of(nm_)
            fam_ms[f_] = fam_ms.get(f_, 0.0) + e0_.elapsed_time(e1_) / n_bd
            fam_calls[f_] = fam_calls.get(f_, 0) + 1
        breakdown = {k_: round(v_, 4) for k_, v_ in sorted(fam_ms.items(), key=lambda kv: -kv[1])}
        breakdown["sum_of_calls"] = round(sum(fam_ms.values()), 4)
        breakdown["step_with_events_and_calibration_launch"] = round(tb, 4)
        breakdown["calls_per_step"] = {k_: v_ // n_bd for k_, v_ in fam_calls.items()}
        breakdown["note"] = ("%d extra steps AFTER the timed region, every C-ABI call bracketed by two HIP events on its stream (weight_gradient "
                             "includes its slab reductions; the steps are drained one by one here, ms_per_step's are not)" % n_bd)
    if rank == 0 and os.environ.get("TNR_BENCH_DUMP_PARAMS"):
        np.save(os.environ["TNR_BENCH_DUMP_PARAMS"], eng.flat[True][::97].float().cpu().numpy())
    dp_info = None
    if use_dp:
        # the data-parallel leg explains itself: bucket sizes in completion order and, per bucket, how long the compute stream
        # stood still behind its all-reduce (events around GradSync.wait_bucket on the stream Engine.step runs on) -- what the
        # overlap with backward did NOT hide.  Max over ranks of the per-step sum.
        ex = gs.exposed_ms()
        per_bucket = [round(sum(ex.get(g, [0.0])) / max(len(ex.get(g, [])), 1), 4) for g in range(len(gs.groups))]
        tot = torch.tensor([sum(per_bucket)], device=dev, dtype=torch.float64)
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        # what stays serial behind the LAST collective on the fp16 path (the per-bucket scans of the earlier buckets hide under
        # the collectives still in flight): 8 more steps with the guard kernels and the update bracketed by events
        tail = None
        if eng.scaler.enabled:
            names = ("tnr_grad_nonfinite_scan", "tnr_grad_nonfinite_commit", "tnr_amsgrad_step_guarded")
            recs = {n_: [] for n_ in names}
            T.TIMED.update(recs)
            n_tail = min(8, K)
            for i in range(W, W + n_tail):
                one_step(eng, gs, i, False)
            torch.cuda.synchronize()
            for n_ in names:
                T.TIMED.pop(n_, None)
            ms_of = lambda n_: sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs[n_]) / n_tail
            nb = max(len(gs.ranges), 1)
            last = sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs[names[0]][nb - 1::nb]) / n_tail if recs[names[0]] else 0.0
            tail = {"scans": round(ms_of(names[0]), 4), "last_scan": round(last, 4), "commit": round(ms_of(names[1]), 4),
                    "amsgrad": round(ms_of(names[2]), 4),
                    "note": "fp16: every bucket is scanned for inf / nan behind its own all-reduce; only the last scan, the commit and the "
                            "guarded update are serial behind the last collective (bf16: no guard, the update runs bucket by bucket)"}
            gs._events = []
        dp_info = {"backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                   "fp16_tail_ms_per_step": tail,
                   "rccl_version": rccl, "algo": gs.algo, "wgrad_units_per_workgroup": wgrad_units, "buckets": a.dp_buckets,
                   "cus_for_persistent_gemms": (autotune or {}).get("chosen", {}).get("cus", 0) or "all",
                   "autotune": autotune,
                   "env": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                                          "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY")},
                   "collectives_mb_in_launch_order": [round(x / 1e6, 2) for x in gs.collective_bytes()],
                   "bucket_mb_in_completion_order": [round(x / 1e6, 2) for x in gs.bucket_bytes()],
                   "exposed_allreduce_ms_per_step_by_collective_rank0": per_bucket,
                   "exposed_allreduce_ms_per_step_max_over_ranks": round(float(D.all_reduce_max(tot).item()), 4),
                   "note": "all-reduce(sum) of fp32 gradients, 1/world folded into AMSGrad; every bucket but the last is launched "
                           "while backward still runs"}
    loss = float(eng.total_loss().item())
    if use_dp and world == 1 and not any(kv.startswith("cus=") for kv in a.gemm_opt):
        # world-1 preflight for the first multi-GPU lease: what it costs THIS step to size the persistent GEMM grids for 248 / 240
        # of the 256 CUs (the CUs RCCL's kernels would get), 5 + 30 steps each, interleaved with the full-chip setting
        K_, W_ = K, W
        K, W = min(30, K_), min(5, W_)
        ms_cu = {}
        for n_cu in (0, 248, 240, 0):
            T.lib().tnr_gemm_set_option(b"cus", n_cu)
            ms_cu.setdefault(n_cu or 256, []).append(1e3 * timed_loop(eng, gs, False) / K)
        T.lib().tnr_gemm_set_option(b"cus", 0)
        K, W = K_, W_
        base = sum(ms_cu[256]) / len(ms_cu[256])
        dp_info["cus_reserved_cost_ms"] = {"all_256": round(base, 4), "248": round(ms_cu[248][0] - base, 4), "240": round(ms_cu[240][0] - base, 4),
                                           "note": "ms per step added by sizing the persistent GEMM grids (tnr_gemm_set_option \"cus\") for fewer CUs at "
                                                   "world size 1, 5 + 30 steps each; with more than one rank the warm-up autotune weighs the same setting "
                                                   "against what the collectives gain (dp.autotune)"}
    if use_dp and a.dp_sweep:
        # the knobs the first multi-GPU run should weigh, each as its own 5 + 30 steps (max over ranks), headline configuration first
        sweep = []
        K_, W_ = K, W
        for units in (1, 2):
            for nb in (5, 3):
                for algo in ("allreduce", "rs_ag"):
                    try:
                        eng = gs = None
                        torch.cuda.empty_cache()
                        E.Engine.WGRAD_UNITS = units
                        e_ = E.Engine(cfg, dev, max_batch=B, dtype=a.dtype)
                        e_.load_state_dict(init_sd)
                        e_.refresh_shadows(all_layers=True)
                        g_ = D.GradSync(e_.flat_g, e_.bucket_ranges(), world, force=a.force_dp, timing=True, merge_layers=nb == 3, algo=algo)
                        K, W = min(30, K_), min(5, W_)
                        d_ = timed_loop(e_, g_, False)
                        ex_ = g_.exposed_ms()
                        sweep.append({"wgrad_units": units, "buckets": nb, "algo": g_.algo, "ms_per_step": round(1e3 * d_ / K, 4),
                                      "value": round(world * B * K / d_, 2),
                                      "exposed_allreduce_ms_per_step_rank0": round(sum(sum(v) for v in ex_.values()) / K, 4)})
                        del e_, g_
                    except Exception as ex:
                        sweep.append({"wgrad_units": units, "buckets": nb, "algo": algo, "error": repr(ex)[:200]})
        K, W = K_, W_
        E.Engine.WGRAD_UNITS = wgrad_units
        dp_info["sweep"] = sweep
    routes = {}
    if rec:
        for _, _, w, shape in rec:
            routes[shape] = T.query("tnr_gemm_nt_route" + ("_f16" if a.dtype == "fp16" else ""), *shape)
    other = None
    if not a.no_other_dtype and a.dedup != "only":
        del eng, gs
        torch.cuda.empty_cache()
        od = "bf16" if a.dtype == "fp16" else "fp16"
        eng2, gs2 = build(od)
        dt2 = timed_loop(eng2, gs2, False)
        other = {"dtype": od, "value": round(world * B * K / dt2, 2), "ms_per_step": round(1e3 * dt2 / K, 4),
                 "final_loss": round(float(eng2.total_loss().item()), 5),
                 "note": "same step with %s activations / weight copies; logits within %s of the fp32 reference "
                         "(tests/test_engine_gpu.py)" % (od, "1.6e-2" if od == "bf16" else "1e-3")}
        del eng2, gs2

    bigger = None
    if world == 1 and not a.no_larger_batch and a.dedup != "only":
        try:
            # beside the headline (BASELINE's B = 32 per GPU): the same step at 4 x the batch, which 288 GB of HBM hold many times over -
            # what the fixed per-launch and per-tile costs are worth (20 timed steps, every NT launch bracketed by events)
            eng = gs = None
            torch.cuda.empty_cache()
            B2, W2, K2 = 4 * B, 3, 20
            h2, m2, c2, l2 = [torch.from_numpy(x).to(dev) for x in synth.impressions(seed + 101, (K2 + W2) * B2, N_NEWS, cfg.U, cfg.C)]
            eng3 = E.Engine(cfg, dev, max_batch=B2, dtype=a.dtype)
            eng3.load_state_dict(init_sd)
            eng3.refresh_shadows(all_layers=True)

            def step3(i):
                s3 = slice(i * B2, (i + 1) * B2)
                eng3.forward_indexed(comb, h2[s3], m2[s3], c2[s3], l2[s3], tables if a.teachers else None)
                eng3.backward()
                eng3.step(lr=1e-4)
            for i in range(W2):
                step3(i)
            rec3 = []
            T.TIMED[TKEY] = rec3
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for i in range(W2, W2 + K2):
                step3(i)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t3
            T.TIMED.pop(TKEY, None)
            ms3, fl3 = sum(e0.elapsed_time(e1) for e0, e1, _, _ in rec3), sum(w for _, _, w, _ in rec3)
            bigger = {"batch_per_gpu": B2, "value": round(B2 * K2 / dt3, 2), "ms_per_step": round(1e3 * dt3 / K2, 4), "steps": K2,
                      "nt_gemm_tflops": round(fl3 / (ms3 * 1e-3) / 1e12, 2), "nt_gemm_frac": round(fl3 / (ms3 * 1e-3) / PEAK_BF16, 4),
                      "note": "not the headline: BASELINE's configuration is 32 impressions per GPU and step (demo.sh:8)"}
            del eng3
        except Exception as ex:                      # an informational leg must never cost the headline line
            T.TIMED.pop(TKEY, None)
            bigger = {"error": repr(ex)[:300]}

    if world == 1 and not a.no_configs and a.dedup != "only":
        eng = gs = None
        torch.cuda.empty_cache()
        for name in LEGS:
            LEGS[name]()
    else:
        legs = None

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
            ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in rec)
            fl = sum(w for _, _, w, _ in rec)
            ach = fl / (ms * 1e-3) / 1e12
            # HBM traffic per launch comes from a separate rocprofv3 --pmc pass (tools/make_profiles.sh); it is only quoted
            # while the library's SOURCES are the ones that were profiled (tnr_hip.source_sha16: a rebuild does not break the link)
            traffic, busy = None, None
            import glob
            for pj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_nt_pmc.json")), reverse=True):
                pm = json.load(open(pj))
                if pm.get("src_sha16") == T.source_sha16() and pm.get("dtype") == a.dtype:
                    traffic = pm.get("hbm_bytes_per_launch")
                    busy = pm.get("mfma_busy_frac")          # SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x GRBM_GUI_ACTIVE): SURVEY 8-d's "MFMA utilisation"
                    break
            names = {128: "gemm_nt_kernel(128x128)", 2128: "gemm_nt256_kernel(256x128)", 256: "gemm_nt_pp_kernel<8,*>(256- / 224-row tiles)",
                     224: "gemm_nt_pp_kernel<7,*>(224- / 192-row tiles)"}
            out["roofline"] = {"bound": "mfma",
                               "kernel": "NT GEMM family (%s MFMA 16x16x32, every forward / dgrad Linear incl. fused epilogues): %s"
                                         % (a.dtype, ", ".join(sorted({names.get(r, str(r)) for r in routes.values()}))),
                               "achieved": round(ach, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                               "frac": round(ach * 1e12 / PEAK_BF16, 4), "traffic": traffic, "mfma_busy_pmc": busy,
                               "launches": len(rec), "launches_timed_every_nth_step": EVENT_EVERY,
                               "avg_launch_us": round(1e3 * ms / len(rec), 2),
                               "algorithmic_flops_per_launch": fl / len(rec)}
        else:
            out["roofline"] = None
        out["parity"] = PARITY.get("headline")       # what the headline model's parity test allows and measured (tools/parity_measured.py):
        #                                              B = 2 golden + `b32` (the benchmark's own batch) + `trajectory` (50 reference steps)
        out["quality"] = PARITY.get("quality")       # tests/test_quality_gpu.py: AUC / MRR / nDCG against the reference-trained golden
        if box is not None:
            out["box"] = box
            cal = box["calibration_launches"]["mfma_heavy"]["us"]
            out["box"]["headline_normalised"] = {
                "reference_mfma_heavy_calibration_us": CAL_REF_US, "value_at_reference_calibration": round(value * cal / CAL_REF_US, 2),
                "note": "value x (this box's MFMA-heavy calibration us / %.1f us, the round-6 build leases' median): first-order only - "
                        "compare RAW values between boxes whose calibration agrees" % CAL_REF_US}
            out["step_breakdown_ms"] = breakdown
        if dp_info is not None:
            out["dp"] = dp_info
        if other is not None:
            out["other_dtype"] = other
        if bigger is not None:
            out["larger_batch"] = bigger
        if legs:
            out["configs"] = legs
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
