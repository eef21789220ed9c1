"""Where a K tile of the ping-pong NT kernel (gemm_nt_pp_kernel) spends its cycles - the barrier-arrival stamps of tools/tn_stamps.py
on the forward / dgrad GEMM; GPU box, probe build:
    tools/probes/build.sh _ntst -DTNR_NT_STAMPS
    N=3072 K=768 FLAGS=1 python tools/nt_stamps.py
Intervals of a K tile (group 0's numbering; group 1 runs one barrier behind): L0 M0 L1 M1 L2 M2 L3 M3, 16 MFMAs per M."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
T.LIB_PATH = os.path.join(ROOT, os.environ.get("LIB", "tools/_ntst"), "libtnr_hip.so")
dev, M = "cuda:0", int(os.environ.get("M", 52800))
N, K, fl = int(os.environ.get("N", 3072)), int(os.environ.get("K", 768)), int(os.environ.get("FLAGS", 1))
td, sfx = torch.float16, "_f16"
a = (torch.randn((M, K), device=dev) * 0.5).to(td); b = (torch.randn((N, K), device=dev) * 0.05).to(td)
c = torch.zeros((M, N), device=dev, dtype=td)
bias = torch.randn(N, device=dev); r = torch.randn((M, N), device=dev).to(td)
L = T.lib()
L.tnr_gemm_set_option(b"pp", int(os.environ.get("PP", 1)))
run = lambda: T.call("tnr_gemm_nt_ex" + sfx, a, K, b, K, c, N, M, N, K, bias, r if fl & 8 else None, N if fl & 8 else 0, None, 0, fl, None)
for _ in range(int(os.environ.get("WARM", 300))): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("C %d x %d, K %d, flags %d: %.1f us per launch (probe build)" % (M, N, K, fl, e0.elapsed_time(e1) * 50))
W, NB = 264, int(os.environ.get("NB", 8))          # NB=4 with PP=2: the register-staged kernel has four barriers per K tile
buf = np.zeros((256, 8, W), np.uint32)
fn = getattr(L, "tnr_debug_nt_stamps" + sfx)
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert fn(buf.ctypes.data, buf.nbytes) == 0
rows = []
for wg in range(256):
    nk = min(int(buf[wg, 0, 4]), 32)
    if nk < 6: continue
    a_ = buf[wg, :, 8:8 + 8 * nk].reshape(8, nk, 8)[:, :, :NB].reshape(8, nk * NB).astype(np.int64)    # 8 slots per K tile, NB used
    a_ = (a_ - a_[0, 0]) & 0xffffffff
    a_[a_ > (1 << 31)] -= (1 << 32)
    n = np.arange(NB + 1, NB * nk - 1)
    arr = np.concatenate([a_[:4][:, n], a_[4:][:, n - 1]])
    arr_prev = np.concatenate([a_[:4][:, n - 1], a_[4:][:, n - 2]])
    rel, rel_prev = arr.max(0), arr_prev.max(0)
    iv, busy = rel - rel_prev, arr - rel_prev
    if iv.min() < 0 or iv.max() > 100000: continue
    per = np.zeros((NB, 9))
    for k in range(NB):
        sel = (n % NB) == k
        per[k, 0] = iv[sel].mean(); per[k, 1:] = busy[:, sel].mean(1)
    rows.append(per)
med = np.median(np.array(rows), axis=0)
print("workgroups analysed: %d" % len(rows))
names0 = ["L0", "M0", "L1", "M1", "L2", "M2", "L3", "M3"][:NB]
names1 = (["M3", "L0", "M0", "L1", "M1", "L2", "M2", "L3"] if NB == 8 else ["M1", "L0", "M0", "L1"])
print("interval  length | group 0: segment, cycles until arrival of waves 0-3 | group 1: segment, waves 4-7")
for k in range(NB):
    print("   %d      %5.0f  |  %s  %5.0f %5.0f %5.0f %5.0f  |  %s  %5.0f %5.0f %5.0f %5.0f" % ((k, med[k, 0], names0[k]) + tuple(med[k, 1:5]) + (names1[k],) + tuple(med[k, 5:9])))
print("sum of intervals: %.0f cycles per K tile" % med[:, 0].sum())
