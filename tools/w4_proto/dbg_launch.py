"""Debug of run_w4_asm.py's launch path (hipModuleLaunchKernel + extra) with a hipcc-compiled kernel, then the assembly kernel's
probe epilogue (one dword per thread) with the K loop skipped."""
import ctypes, os, struct, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import run_w4_asm as R
hip = R.hip
src = '''#include <hip/hip_runtime.h>
extern "C" __global__ void kecho(const void* A, const void* B, unsigned* C, int lda, int ldb, int ldc, int M, int nk, int nbn) {
    unsigned* o = C + (blockIdx.x * 256 + threadIdx.x) * 4;
    o[0] = (unsigned)(size_t)A; o[1] = lda * 1000003u + ldb * 10007u + ldc; o[2] = M * 100u + nk * 10u + nbn; o[3] = blockIdx.x * 256 + threadIdx.x;
}'''
open("/tmp/kecho.hip", "w").write(src)
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "--genco", "-o", "/tmp/kecho.hsaco", "/tmp/kecho.hip"], check=True)
mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
assert hip.hipModuleLoad(ctypes.byref(mod), b"/tmp/kecho.hsaco") == 0
assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"kecho") == 0
M, N, K = 512, 768, 256
a = torch.zeros((M, K), device="cuda", dtype=torch.float16); b = torch.zeros((N, K), device="cuda", dtype=torch.float16)
c = torch.zeros((M, N), device="cuda", dtype=torch.float16)
R.launch(fn, a, b, c, M, N, K)
torch.cuda.synchronize()
o = c.view(torch.int32).reshape(-1)[:8].tolist()
print("echo:", o, "expect A lo", a.data_ptr() & 0xffffffff, (K * 2) * 1000003 + (K * 2) * 10007 + N * 2, M * 100 + (K // 64) * 10 + N // 256, flush=True)
# assembly kernel, K loop skipped, probe epilogue
fn2 = R.build("mfma", "skip=1")
c.fill_(3.0)
R.launch(fn2, a, b, c, M, N, K)
torch.cuda.synchronize()
v = c.view(torch.int32).reshape(-1)
grid = ((M + 255) // 256) * (N // 256)
print("asm probe epilogue: first %d dwords zero: %s ; next dword untouched: %s" % (grid * 256, bool((v[:grid * 256] == 0).all()), int(v[grid * 256]) != 0), flush=True)
