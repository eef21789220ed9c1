// Fused AMSGrad over the flat trainable-parameter buffer + refresh of the bf16 weight copies
// (row-major and transposed) the GEMMs read.  torch.optim.Adam(amsgrad=True) semantics (run.py:134).
#include "common.h"

namespace {

template <bool AMS>
__global__ __launch_bounds__(256) void amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      float* __restrict__ vmax, int64_t n, float lr_c1, float inv_sqrt_c2,
                                                      float b1, float b2, float eps, float gscale) {
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
extern "C" int tnr_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, int step, float lr,
                                float beta1, float beta2, float eps, float grad_scale, void* stream) {
    TNR_CHECK_ARG(p && g && m && v && n >= 1 && step >= 1, "tnr_amsgrad_step: bad argument");     // vmax NULL = plain Adam
    TNR_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                      ((uintptr_t)v % 16) == 0 && ((uintptr_t)vmax % 16) == 0, "tnr_amsgrad_step: 16-byte alignment");
    double c1 = 1.0 - pow((double)beta1, step), c2 = 1.0 - pow((double)beta2, step);
    float lr_c1 = (float)((double)lr / c1), inv_sqrt_c2 = (float)(1.0 / sqrt(c2));
    int64_t nthr = (n + 3) / 4;
    if (vmax)
        hipLaunchKernelGGL(amsgrad_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                           v, vmax, n, lr_c1, inv_sqrt_c2, beta1, beta2, eps, grad_scale);
    else      // plain Adam (Post-train_KD.ipynb cell 18: optim.Adam without amsgrad)
        hipLaunchKernelGGL(amsgrad_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                           v, vmax, n, lr_c1, inv_sqrt_c2, beta1, beta2, eps, grad_scale);
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
