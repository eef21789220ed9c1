"""Long-sequence attention (forward and backward) of two builds of the library on the same seeded operands: outputs saved for a bitwise comparison,
timing printed; GPU box.    LIB=tools/_old/libtnr_hip.so OUT=/tmp/a.pt python tools/scratch/attn_long_bwd_check.py ; (again without LIB,
OUT=/tmp/b.pt) ; python tools/scratch/attn_long_bwd_check.py cmp /tmp/a.pt /tmp/b.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch
if len(sys.argv) > 1 and sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        print(k, "bit-identical" if torch.equal(a[k], b[k]) else "DIFFERENT: max |d| %.3e" % float((a[k].float() - b[k].float()).abs().max()))
    sys.exit(0)
import tnr_hip as T
if os.environ.get("LIB"):
    T.LIB_PATH = os.path.join(ROOT, os.environ["LIB"])
dev, A, H = "cuda:0", 12, 768
out = {}
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for dt, td, sfx in (("fp16", torch.float16, "_f16"), ("bf16", torch.bfloat16, "")):
    for (L, N, drop) in ((512, 32, 0.0), (128, 32, 0.0), (100, 5, 0.0), (70, 3, 0.1), (33, 2, 0.0), (512, 104, 0.0), (128, 416, 0.0)):
        g = torch.Generator(device=dev).manual_seed(L * 7 + N)
        Lr = (L + 31) // 32 * 32
        qkv = (torch.randn(N * L, 3 * H, device=dev, generator=g) * 0.5).to(td)
        mask = torch.zeros(N, Lr, device=dev); mask[:, max(L - 9, 1):L] = -10000.0; mask[:, L:] = -1e30
        rel = torch.randn(A, Lr, Lr, device=dev, generator=g) * 0.1
        ctx = torch.zeros(N * L, H, device=dev, dtype=td); dctx = (torch.randn(N * L, H, device=dev, generator=g) * 0.1).to(td)
        lse = torch.zeros(N, A, Lr, device=dev); delta = torch.full((N, A, Lr), 7.0, device=dev); dqkv = torch.full_like(qkv, 3.0)
        site = None
        if drop:
            site = T.Dropout.site_of(drop, 11, T.DROP_PROB, 1, 0)
        fa = (qkv, mask, rel, ctx, lse, N, L, A)
        ba = (qkv, mask, rel, ctx, dctx, lse, delta, dqkv, N, L, A)
        if site is not None:
            T.call("tnr_attn_long_fwd_do" + sfx, *fa, site); T.call("tnr_attn_long_bwd_do" + sfx, *ba, site)
        else:
            T.call("tnr_attn_long_fwd" + sfx, *fa); T.call("tnr_attn_long_bwd" + sfx, *ba)
        torch.cuda.synchronize()
        key = "%s L=%d N=%d%s" % (dt, L, N, " dropout" if site is not None else "")
        out[key + " dqkv"], out[key + " delta"] = dqkv.cpu(), delta.cpu()
        out[key + " ctx"], out[key + " lse"] = ctx.cpu(), lse.cpu()
        if N >= 32 and dt == "fp16":
            ts = []
            for _ in range(10):
                junk.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); T.call("tnr_attn_long_bwd" + sfx, *ba); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            tf = []
            for _ in range(10):
                junk.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); T.call("tnr_attn_long_fwd" + sfx, *fa); e1.record(); torch.cuda.synchronize()
                tf.append(e0.elapsed_time(e1) * 1e3)
            print("%-24s fwd %.1f us   bwd (dq + dkv) %.1f us" % (key, sorted(tf)[5], sorted(ts)[5]), flush=True)
torch.save(out, os.environ.get("OUT", "/tmp/attn_long_bwd.pt"))
