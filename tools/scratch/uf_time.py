"""Time of the fused user-encoder forward (tnr_user_score_fwd with epre = NULL) at the step's shape; LIB=<build> for probe copies."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
if os.environ.get("LIB"): T.LIB_PATH = os.path.join(ROOT, os.environ["LIB"])
dev = "cuda:0"
B, U, C, D, Q, R = 32, 50, 5, 256, 200, 1792
for nm in (1, 4):
    vec = torch.randn(nm, R, D, device=dev) * 0.3
    hidx = torch.randint(0, R, (B, U), device=dev, dtype=torch.int32); cidx = torch.randint(0, R, (B, C), device=dev, dtype=torch.int32)
    mask = (torch.rand(B, U, device=dev) > 0.3).float()
    pad, w1, b1 = torch.randn(nm, D, device=dev) * 0.1, torch.randn(nm, Q, D, device=dev) * 0.05, torch.randn(nm, Q, device=dev) * 0.1
    w2, b2 = torch.randn(nm, Q, device=dev) * 0.1, torch.zeros(nm, device=dev)
    user, score = torch.zeros(nm, B, D, device=dev), torch.zeros(nm, B, C, device=dev)
    e, alpha, den = torch.zeros(nm, B, U, Q, device=dev), torch.zeros(nm, B, U, device=dev), torch.zeros(nm, B, device=dev)
    run = lambda: T.call("tnr_user_score_fwd", vec, R, hidx, cidx, mask, pad, w1, b1, w2, b2, 0, None, None, user, B * D, score, e, alpha, den, nm, B, U, C, D, Q)
    for _ in range(3): run()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("%s n_model=%d: %.1f us" % (os.environ.get("LIB", "shipped"), nm, sorted(ts)[10]), flush=True)
