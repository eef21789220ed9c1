"""The four weight gradients of a layer: four launches (one shared slab buffer, as the engine used to) against one grouped launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T, engine as E
dev, M, td, sfx = "cuda:0", 52800, torch.float16, "_f16"
H, I = 768, 3072
Mp = (M + 127) // 128 * 128
shapes = [(H, I), (I, H), (H, H), (3 * H, H)]
probs = []
wsmax = torch.zeros(max(E.Engine._wgrad_splits(n, k)[1] for n, k in shapes), device=dev)
for N, K in shapes:
    dy = torch.zeros((Mp, N), device=dev, dtype=td); dy[:M] = (torch.randn((M, N), device=dev) * 0.1).to(td)
    x = torch.zeros((Mp, K), device=dev, dtype=td); x[:M] = torch.randn((M, K), device=dev).to(td)
    sp, el = E.Engine._wgrad_splits(N, K)
    probs.append(dict(dY=dy, lddy=N, X=x, ldx=K, dW=torch.zeros((N, K), device=dev), lddw=K, M=M, N=N, K=K, ws=torch.zeros(el, device=dev),
                      splits=sp, accumulate=0, out_scale=1.0))
def separate():
    for q in probs:
        T.call("tnr_gemm_tn_wgrad_ex" + sfx, q["dY"], q["N"], q["X"], q["K"], q["dW"], q["K"], M, q["N"], q["K"], wsmax, q["splits"], 0, 1.0)
def grouped(): T.wgrad_group(probs, f16=True)
def pairs():
    T.wgrad_group(probs[:2], f16=True); T.wgrad_group(probs[2:], f16=True)
variants = [("separate", separate), ("grouped", grouped), ("two groups of two", pairs)]
for spl in (3, 4, 5, 6, 8):                        # the same number of splits for every member: units of equal length
    pp = [dict(q, splits=spl, ws=torch.zeros(q["N"] * q["K"] * spl, device=dev)) for q in probs]
    variants.append(("grouped, %d splits each" % spl, (lambda a: (lambda: T.wgrad_group(a, f16=True)))(pp)))
res = {}
for rnd in range(5):
    for name, fn in variants:
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) * 200)
for k, v in res.items(): print("%-20s %.1f us" % (k, sorted(v)[2]))
