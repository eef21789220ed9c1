// Fused AMSGrad over the flat trainable-parameter buffer + refresh of the bf16 weight copies
// (row-major and transposed) the GEMMs read.  torch.optim.Adam(amsgrad=True) semantics (run.py:134).
#include <algorithm>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// guard (may be NULL): guard[0] = stamp of the last optimiser step whose gradient held a non-finite value, guard[1] = how many
// steps have been skipped for that so far (tnr_grad_nonfinite).  A launch stamped guard[0] leaves parameters and state untouched:
// the skipped step of dynamic loss scaling.  Adam's bias corrections depend on the number of steps actually TAKEN, which the host
// knows only with a lag: it passes the factors for step, step - 1, step - 2 and how many skips it has accounted for, the kernel
// picks by the skips it has not (all factors computed on the host, in double: the same bits as the unguarded launch).
struct AdamFactors { float lr_c1[3], inv_sqrt_c2[3]; unsigned known_skips; };
template <bool AMS>
__global__ __launch_bounds__(256) void amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      float* __restrict__ vmax, int64_t n, AdamFactors f,
                                                      float b1, float b2, float eps, float gscale,
                                                      const unsigned* __restrict__ guard, unsigned stamp) {
    float lr_c1 = f.lr_c1[0], inv_sqrt_c2 = f.inv_sqrt_c2[0];
    if (guard) {
        if (guard[0] == stamp) return;
        const unsigned k = guard[1] - f.known_skips;
        if (k == 1) { lr_c1 = f.lr_c1[1]; inv_sqrt_c2 = f.inv_sqrt_c2[1]; }
        else if (k >= 2) { lr_c1 = f.lr_c1[2]; inv_sqrt_c2 = f.inv_sqrt_c2[2]; }
    }
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        f32x4 gv = *(const f32x4*)(g + i) * gscale;
        f32x4 mv = *(const f32x4*)(m + i) * b1 + (1.f - b1) * gv;
        f32x4 vv = *(const f32x4*)(v + i) * b2 + (1.f - b2) * gv * gv;
        f32x4 vm = vv;
        if (AMS) vm = *(const f32x4*)(vmax + i);
        f32x4 pv = *(const f32x4*)(p + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (AMS) vm[r] = fmaxf(vm[r], vv[r]);
            pv[r] -= lr_c1 * (mv[r] / (sqrtf(vm[r]) * inv_sqrt_c2 + eps));
        }
        *(f32x4*)(m + i) = mv;
        *(f32x4*)(v + i) = vv;
        if (AMS) *(f32x4*)(vmax + i) = vm;
        *(f32x4*)(p + i) = pv;
    } else {
        for (; i < n; ++i) {
            float gv = g[i] * gscale;
            float mv = m[i] * b1 + (1.f - b1) * gv;
            float vv = v[i] * b2 + (1.f - b2) * gv * gv;
            float vm = AMS ? fmaxf(vmax[i], vv) : vv;
            m[i] = mv; v[i] = vv;
            if (AMS) vmax[i] = vm;
            p[i] -= lr_c1 * (mv / (sqrtf(vm) * inv_sqrt_c2 + eps));
        }
    }
}

// any inf / nan among g[0 .. n)?  -> guard[0] = max(guard[0], stamp): one atomic per OFFENDING workgroup, none on a clean gradient
// (an arrival counter on one address cost more than the scan: 2 048 same-address atomics = 30-70 us).  The skipped step is counted
// in guard[1] by the one-thread kernel behind it.  16-byte reads, four in flight per lane.
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* __restrict__ g, int64_t n, unsigned* guard, unsigned stamp) {
    const int64_t stride = (int64_t)gridDim.x * 4096;
    unsigned bad = 0;
    for (int64_t base = (int64_t)blockIdx.x * 4096; base < n; base += stride) {
        const int64_t i0 = base + threadIdx.x * 4;
        if (base + 4096 <= n) {
            u32x4_t b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = *(const u32x4_t*)(g + i0 + u * 1024);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) bad |= ((b[u][r] & 0x7F800000u) == 0x7F800000u);
        } else {
            for (int u = 0; u < 4; ++u)
                for (int64_t j = i0 + u * 1024; j < i0 + u * 1024 + 4 && j < n; ++j)
                    bad |= ((__float_as_uint(g[j]) & 0x7F800000u) == 0x7F800000u);
        }
    }
    const int any = __syncthreads_or((int)bad);
    if (any && threadIdx.x == 0) __hip_atomic_fetch_max(guard, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void grad_nonfinite_count_kernel(unsigned* guard, unsigned stamp) {      // <<<1, 1>>>, stream-ordered behind the scan
    if (guard[0] == stamp) guard[1] += 1;
}

// one workgroup per 32x32 tile of some weight matrix (descriptor table on device)
__global__ __launch_bounds__(256) void refresh_kernel(const int64_t* __restrict__ desc, int n_desc,
                                                      const int64_t* __restrict__ tile_start) {
    __shared__ float t[32][33];
    int64_t tile = blockIdx.x;
    int di = 0;
    while (di + 1 < n_desc && tile >= tile_start[di + 1]) ++di;
    const int64_t* d = desc + (int64_t)di * 8;
    const float* src = (const float*)d[0];
    const int64_t rows = d[1], cols = d[2];
    bf16* dst = (bf16*)d[3];
    const int64_t ld = d[4];
    bf16* dstT = (bf16*)d[5];
    const int64_t ldT = d[6];
    const int64_t local = tile - tile_start[di];
    const int64_t tcols = (cols + 31) / 32;
    const int64_t r0 = (local / tcols) * 32, c0 = (local % tcols) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int64_t r = r0 + ty + 8 * k, c = c0 + tx;
        float v = (r < rows && c < cols) ? src[r * cols + c] : 0.f;
        t[ty + 8 * k][tx] = v;
        if (dst && r < rows && c < cols) dst[r * ld + c] = (bf16)v;
    }
    __syncthreads();
    if (dstT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int64_t c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < rows && c < cols) dstT[c * ldT + r] = (bf16)t[tx][ty + 8 * k];
        }
    }
}

}  // namespace

#ifndef TNR_BUILD_F16
extern "C" int tnr_amsgrad_step_guarded(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, int step, float lr,
                                        float beta1, float beta2, float eps, float grad_scale, const unsigned* guard,
                                        unsigned stamp, unsigned known_skips, void* stream);
extern "C" int tnr_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, int step, float lr,
                                float beta1, float beta2, float eps, float grad_scale, void* stream) {
    return tnr_amsgrad_step_guarded(p, g, m, v, vmax, n, step, lr, beta1, beta2, eps, grad_scale, nullptr, 0u, 0u, stream);
}

extern "C" int tnr_grad_nonfinite_scan(const float* g, int64_t n, unsigned* guard, unsigned stamp, void* stream) {
    TNR_CHECK_ARG(g && guard && n >= 1 && stamp >= 1 && ((uintptr_t)g % 16) == 0, "tnr_grad_nonfinite_scan: bad argument");
    const unsigned grid = (unsigned)std::min<int64_t>((n + 4095) / 4096, 2048);
    hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, guard, stamp);
    TNR_CHECK_LAUNCH("tnr_grad_nonfinite_scan");
    return TNR_OK;
}

extern "C" int tnr_grad_nonfinite_commit(unsigned* guard, unsigned stamp, void* stream) {
    TNR_CHECK_ARG(guard && stamp >= 1, "tnr_grad_nonfinite_commit: bad argument");
    hipLaunchKernelGGL(grad_nonfinite_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, guard, stamp);
    TNR_CHECK_LAUNCH("tnr_grad_nonfinite_commit");
    return TNR_OK;
}

extern "C" int tnr_grad_nonfinite(const float* g, int64_t n, unsigned* guard, unsigned stamp, void* stream) {
    TNR_CHECK_ARG(g && guard && n >= 1 && stamp >= 1 && ((uintptr_t)g % 16) == 0, "tnr_grad_nonfinite: bad argument");
    const unsigned grid = (unsigned)std::min<int64_t>((n + 4095) / 4096, 2048);
    hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, guard, stamp);
    hipLaunchKernelGGL(grad_nonfinite_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, guard, stamp);
    TNR_CHECK_LAUNCH("tnr_grad_nonfinite");
    return TNR_OK;
}

extern "C" int tnr_amsgrad_step_guarded(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, int step, float lr,
                                        float beta1, float beta2, float eps, float grad_scale, const unsigned* guard,
                                        unsigned stamp, unsigned known_skips, void* stream) {
    TNR_CHECK_ARG(p && g && m && v && n >= 1 && step >= 1, "tnr_amsgrad_step: bad argument");     // vmax NULL = plain Adam
    TNR_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                      ((uintptr_t)v % 16) == 0 && ((uintptr_t)vmax % 16) == 0, "tnr_amsgrad_step: 16-byte alignment");
    AdamFactors f;
    for (int k = 0; k < 3; ++k) {
        const int t = step - k >= 1 ? step - k : 1;
        const double c1 = 1.0 - pow((double)beta1, t), c2 = 1.0 - pow((double)beta2, t);
        f.lr_c1[k] = (float)((double)lr / c1);
        f.inv_sqrt_c2[k] = (float)(1.0 / sqrt(c2));
    }
    f.known_skips = known_skips;
    int64_t nthr = (n + 3) / 4;
    if (vmax)
        hipLaunchKernelGGL(amsgrad_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                           v, vmax, n, f, beta1, beta2, eps, grad_scale, guard, stamp);
    else      // plain Adam (Post-train_KD.ipynb cell 18: optim.Adam without amsgrad)
        hipLaunchKernelGGL(amsgrad_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                           v, vmax, n, f, beta1, beta2, eps, grad_scale, guard, stamp);
    TNR_CHECK_LAUNCH("tnr_amsgrad_step");
    return TNR_OK;
}

#endif

extern "C" int TNR_NAME(tnr_refresh_shadows)(const int64_t* desc, int n_desc, int64_t total_tiles, const int64_t* tile_start,
                                   void* stream) {
    TNR_CHECK_ARG(desc && tile_start && n_desc >= 1 && total_tiles >= 1, "tnr_refresh_shadows: bad argument");
    hipLaunchKernelGGL(refresh_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, desc, n_desc,
                       tile_start);
    TNR_CHECK_LAUNCH("tnr_refresh_shadows");
    return TNR_OK;
}
