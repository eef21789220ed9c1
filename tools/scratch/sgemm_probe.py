"""What the heads' grouped fp32 GEMM launches wait for: the step's three groups timed on probe copies of the library
(make -C tiny-newsrec_amd/csrc BUILD=../../tools/_sgN EXTRA=-DTNR_SG_SKIP=N: 1 no MFMAs, 2 no global loads after the first K step); GPU box.
    LIB=tools/_sg1 python tools/scratch/sgemm_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
if os.environ.get("LIB"):
    T.LIB_PATH = os.path.join(ROOT, os.environ["LIB"], "libtnr_hip.so")
import engine as E
dev = "cuda:0"
cfg = E.EngineConfig(n_layers=4, trainable_layers=(2, 3), num_teachers=4)
eng = E.Engine(cfg, dev, max_batch=32, dtype="fp16")
B, U, C, D, H, T_, Qu = 32, cfg.U, cfg.C, cfg.D, cfg.H, cfg.T, cfg.Qu
N, Rt = B * (U + C), B * (U + C) + B
f = lambda *s: torch.randn(s, device=dev)
nv, S, X, Pm, dP, dvec, dnv = f(N, H), f(Rt, D), f(T_, Rt, D), f(T_, Rt, D), f(T_, Rt, D), f(N, D), f(N, H)
Wt, bt, wd, dwd, dWt = f(T_, D, D), f(T_, D), f(D, H), f(D, H), f(T_, D, D)
dpre, hv, w1, dw1, dhv = f(B * U, Qu), f(B * U, D), f(Qu, D), f(Qu, D), f(B * U, D)
P = eng._sgemm_problem
groups = {
 "forward: dense + teacher projections": lambda: [P(nv, H, 1, 0, wd, H, 1, 0, S, D, 0, None, 0, N, D, H), P(X, D, 1, X.stride(0), Wt, D, 1, D * D, Pm, D, Pm.stride(0), bt, D, Rt, D, D, batch=T_)],
 "backward heads: transform grads + user dW1 + dhv": lambda: [P(dP, 1, D, Rt * D, X, 1, D, X.stride(0), dWt, D, D * D, None, 0, D, D, Rt, batch=T_, ksplit=eng.KS),
      P(dpre, 1, Qu, 0, hv, 1, D, 0, dw1, D, 0, None, 0, Qu, D, B * U, ksplit=eng.KS, part=eng.sg_part2), P(dpre, Qu, 1, 0, w1, 1, D, 0, dhv, D, 0, None, 0, B * U, D, Qu)],
 "backward encoder: dense.weight grad + dnv": lambda: [P(dvec, 1, D, 0, nv, 1, H, 0, dwd, H, 0, None, 0, D, H, N, ksplit=eng.KS), P(dvec, D, 1, 0, wd, 1, H, 0, dnv, H, 0, None, 0, N, H, D)],
}
for name, mk in groups.items():
    for _ in range(3): eng._sgemm_group(mk())
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): eng._sgemm_group(mk())
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 200)
    print("%-52s %6.1f us per launch pair (GEMM + split reduce)" % (name, sorted(ts)[7]), flush=True)
