"""Where an m step of the register-staged weight-gradient kernel (gemm_tn_rs_kernel) spends its cycles; GPU box, probe build:
    tools/probes/build.sh _tnst -DTNR_TN_STAMPS
    LIB=tools/_tnst python tools/tn_stamps.py            (N=3072 K=768 by default)
Every wave stamps s_memtime on ARRIVAL at each of the 4 barriers of m steps 32-95.  Group 1 (waves 4-7) runs one barrier behind, so
barrier n of group 0 is barrier n - 1 of group 1; release = the latest arrival.  Printed, as medians over the workgroups of means
over the steps: the interval between two releases, and how long each wave worked in it before it arrived (the rest of the
interval it waited for the others).  Clock = cycles / s_memrealtime (100 MHz) over the unit."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, tnr_hip as T
import engine as E
T.LIB_PATH = os.path.join(ROOT, os.environ.get("LIB", "tools/_tnst"), "libtnr_hip.so")
dev, M = "cuda:0", int(os.environ.get("M", 52800))
N, K = int(os.environ.get("N", 3072)), int(os.environ.get("K", 768))
Mp = (M + 127) // 128 * 128
td, sfx = torch.float16, "_f16"
dy = torch.zeros((Mp, N), device=dev, dtype=td); dy[:M] = (torch.randn((M, N), device=dev) * 0.1).to(td)
x = torch.zeros((Mp, K), device=dev, dtype=td); x[:M] = torch.randn((M, K), device=dev).to(td)
dw = torch.zeros((N, K), device=dev)
sp = E.Engine._wgrad_splits(N, K)[0]
ws = torch.zeros(T.query("tnr_gemm_tn_ws_elems" + sfx, N, K, sp), device=dev)
L = T.lib()
L.tnr_gemm_set_option(b"tnpp", 2)
run = lambda: T.call("tnr_gemm_tn_wgrad" + sfx, dy, N, x, K, dw, K, M, N, K, ws, sp, 0)
for _ in range(int(os.environ.get("WARM", 300))): run()          # the chip at its loaded clock
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("dW %d x %d, %d splits: %.1f us per launch incl. slab reduce (probe build)" % (N, K, sp, e0.elapsed_time(e1) * 50))
W, T0, NT, NB = 264, 32, 64, 4
buf = np.zeros((256, 8, W), np.uint32)
fn = getattr(L, "tnr_debug_tn_stamps" + sfx)
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert fn(buf.ctypes.data, buf.nbytes) == 0
rows, clk = [], []
d = lambda a, b: ((int(a) - int(b)) & 0xffffffff)
for wg in range(256):
    nk = int(buf[wg, 0, 4])
    if nk < T0 + NT + 2: continue
    cyc, rt = d(buf[wg, 0, 2], buf[wg, 0, 0]), d(buf[wg, 0, 3], buf[wg, 0, 1])
    clk.append((cyc / max(rt, 1) * 100.0, cyc / nk))
    a = buf[wg, :, 8:8 + NB * NT].astype(np.int64)           # [wave][n], n = NB (t - 32) + k in group 0's numbering
    a = (a - a[0, 0]) & 0xffffffff                           # low 32 bits, relative
    a[a > (1 << 31)] -= (1 << 32)
    n = np.arange(NB + 1, NB * NT - 1)
    arr = np.concatenate([a[:4][:, n], a[4:][:, n - 1]])     # group 1 runs one barrier behind
    arr_prev = np.concatenate([a[:4][:, n - 1], a[4:][:, n - 2]])
    rel, rel_prev = arr.max(0), arr_prev.max(0)
    iv = rel - rel_prev
    busy = arr - rel_prev
    if iv.min() < 0 or iv.max() > 100000: continue
    per = np.zeros((NB, 9))
    for k in range(NB):
        sel = (n % NB) == k
        per[k, 0] = iv[sel].mean()
        per[k, 1:] = busy[:, sel].mean(1)
    rows.append(per)
rows, clk = np.array(rows), np.array(clk)
print("workgroups analysed: %d ; clock %.0f MHz (median), %.0f cycles per m step over the whole unit" % (
    len(rows), np.median(clk[:, 0]), np.median(clk[:, 1])))
med = np.median(rows, axis=0)
names0 = ["L_A", "M_A", "L_B", "M_B"]
names1 = ["M_B", "L_A", "M_A", "L_B"]
print("interval  length | group 0: segment, cycles until arrival of waves 0-3 | group 1: segment, waves 4-7")
for k in range(NB):
    print("   %d      %5.0f  |  %s  %5.0f %5.0f %5.0f %5.0f  |  %s  %5.0f %5.0f %5.0f %5.0f" % ((k, med[k, 0], names0[k]) + tuple(med[k, 1:5]) + (names1[k],) + tuple(med[k, 5:9])))
print("sum of intervals: %.0f cycles per m step" % med[:, 0].sum())
