// Shared device/host helpers for libtnr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tnr_hip.h"

// The 16-bit activation type.  The sources are compiled twice: once with bf16 (entry points tnr_*) and once
// with -DTNR_F16 (IEEE half, entry points tnr_*_f16, same MFMA rate, 3 more mantissa bits).  The identifier
// `bf16` below means "the 16-bit type of this build".
#ifdef TNR_BUILD_F16
typedef _Float16 bf16;
#define TNR_NAME(x) x##_f16
#define TNR_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define TNR_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 bf16;
#define TNR_NAME(x) x
#define TNR_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define TNR_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
typedef __attribute__((ext_vector_type(8))) bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define WAVE 64

void tnr_set_error(const char* fmt, ...);

// Process-wide GEMM options (api.cpp).  Defaults are the product configuration; only tools/ change them, through
// tnr_gemm_set_option(), for same-process A/B runs.  The library never reads environment variables.
struct TnrGemmOpts {
    int ver;         // 3 = route by shape (default) ; 1 / 2 = force the 128x128 / 256x128 kernels
    int gm;          // rasterisation group height in row tiles
    int fine_pct;    // 256x256 grid fill (percent of the CUs) below which the 128x128 kernel is used
    int allow_fine;  // 0 = never fall back to the 128x128 kernel for sparse grids
    int bm;          // 0 = pick the tile height per launch ; 224 / 256 = force it
    int nt;          // 1 = non-temporal accesses for once-touched epilogue operands
    int pp;          // 1 = ping-pong main loop (two wave groups staggered by a barrier), 0 = plain two-buffer loop
    int tnpp;        // weight gradient: non-zero (default 2) = register-staged persistent ping-pong loop, 0 = plain two-buffer loop
    int mix;         // ping-pong NT kernel: 1 = row panels of two heights so that the tiles fill whole rounds, 0 = one height
    int cus;         // 0 = plan and size the persistent GEMM grids for the device's CUs ; n = for n of them (two kernels side by side)
    void* clock_buf; // tnr_gemm_clock_stamps: device buffer of clock_n (cycles, 100 MHz ticks) pairs the persistent NT kernel fills, or NULL
    int clock_n;
};
TnrGemmOpts* tnr_gemm_opts();
// The tile-queue counter set of (current device, stream) for the persistent GEMM kernels of BOTH builds (defined once, in the
// bf16 build of gemm.hip); reset = zero it again (stream-ordered).  NULL + tnr_last_error when the table is full.
unsigned* tnr_pp_queue_of(void* stream, bool reset);

// Runs `body` once per device of this process (function attributes such as the dynamic LDS limit are set per device); two
// threads racing through it both run the idempotent body.
#define TNR_ONCE_PER_DEVICE(body)                                                       \
    do {                                                                                \
        static std::atomic<unsigned long long> tnr_done_{0};                            \
        int tnr_dev_ = 0;                                                               \
        (void)hipGetDevice(&tnr_dev_);                                                  \
        const unsigned long long tnr_bit_ = 1ull << (tnr_dev_ & 63);                    \
        if (!(tnr_done_.load(std::memory_order_acquire) & tnr_bit_)) {                  \
            body;                                                                       \
            tnr_done_.fetch_or(tnr_bit_, std::memory_order_release);                    \
        }                                                                               \
    } while (0)

#define TNR_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            tnr_set_error(__VA_ARGS__);   \
            return TNR_EINVAL;            \
        }                                 \
    } while (0)

// public dropout descriptor (host memory, may be NULL = off) -> kernel argument
#include "dropout.h"
static inline int tnr_make_drop(const tnr_dropout_t* d, TnrDrop* o, const char* who) {
    *o = TnrDrop{0u, 0u, 0u, 0u, 0u, 1.0f};
    if (!d || d->p <= 0.0) return TNR_OK;
    if (!(d->p < 1.0)) { tnr_set_error("%s: dropout p must be in [0, 1)", who); return TNR_EINVAL; }
    o->k0 = (uint32_t)d->seed;
    o->k1 = (uint32_t)(d->seed >> 32);
    o->site = d->site;
    o->call = d->call;
    o->thresh = (uint32_t)(d->p * 65536.0 + 0.5);
    // p below 2^-17 rounds to "keep everything": then nothing is scaled either (forward kernels test thresh, backward ones
    // multiply by scale -- both must see the same effective dropout)
    o->scale = o->thresh ? (float)(1.0 / (1.0 - d->p)) : 1.0f;
    return TNR_OK;
}

#define TNR_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            tnr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return TNR_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 rounding of the outputs):
// one v_rcp + one v_exp + 6 FMAs instead of libm erff's branches.  e = exp(-z*z) is returned for reuse.
// tanh for the additive-attention pre-activations (model_bert.py:25, :37): 1 - 2 / (e^2x + 1) on the hardware exp2 / rcp, an odd
// polynomial below |x| = 0.06 where that form cancels.  |error| <= 3e-7 (libm's tanhf: ~1e-7, at ~60 instructions with divergent
// branches - it made the tanh epilogue of the pooling GEMM cost as much as its K loop and was a third of the user-encoder
// kernel).  ONE function for every kernel: a news vector must not depend on which tile kernel encoded it.
__device__ __forceinline__ float tnr_tanh(float x) {
    const float x2 = x * x;
    const float p = x * (1.f - x2 * (0.33333334f - x2 * 0.13333334f));
    const float t = 1.f - 2.f * __frcp_rn(__expf(2.f * x) + 1.f);
    return fabsf(x) < 0.06f ? p : t;
}

__device__ __forceinline__ float erf_as(float z, float& e) {
    float a = fabsf(z);
    float t = __frcp_rn(1.0f + 0.3275911f * a);
    e = __expf(-a * a);
    float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    float r = 1.0f - p * e;
    return z < 0.f ? -r : r;
}
// erf-GELU of transformers' BertIntermediate (call site tnlrv3/modeling.py:305) and its derivative
__device__ __forceinline__ float gelu_erf(float x) {
    float e;
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f, e));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float e;                                  // e = exp(-x*x/2): also the Gaussian pdf's exponential
    float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f, e));
    return cdf + x * e * 0.39894228040143268f;
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// transposed LDS read: per 16-lane group a 4-row x 16-col block of 16-bit elements, delivered
// column-major (lane i gets column i, rows 0..3); lane 4q+p supplies the address of row q, cols 4p..4p+3.
__device__ __forceinline__ bf16x4 ds_read_tr16(const void* lds_ptr) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lds_ptr));
}
