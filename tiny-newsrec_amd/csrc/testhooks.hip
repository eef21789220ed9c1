// libtnr_testhooks.so -- NOT part of the product library (libtnr_hip.so exports no test hooks): tests/ and tools/ load it beside it.
#include "common.h"

// n_wg workgroups that each take a whole CU (all of its LDS) and spin for `us` microseconds on `stream` -- a stand-in for a
// communication kernel holding CUs while the step's GEMMs run on another stream (tests/test_bench_shapes_gpu.py runs the
// persistent GEMMs beside it: late-starting workgroups, tiles taken over by the others; tools/cu_contention.py times it)
namespace {
__global__ __launch_bounds__(1024) void cu_hog_kernel(unsigned long long ticks) {
    extern __shared__ char hog_lds[];
    if (threadIdx.x == 0) hog_lds[0] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace

void tnr_set_error(const char*, ...) {}

extern "C" int tnr_debug_cu_hog(int n_wg, int us, void* stream) {
    if (!(n_wg >= 1 && n_wg <= 1024 && us >= 0 && us <= 100000)) return TNR_EINVAL;       // 1..1024 workgroups, at most 100 ms
    TNR_ONCE_PER_DEVICE({ (void)hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(cu_hog_kernel, dim3((unsigned)n_wg), dim3(1024), 160 * 1024, (hipStream_t)stream, (unsigned long long)us * 100ull);
    return hipGetLastError() == hipSuccess ? TNR_OK : TNR_ELAUNCH;
}
