// Heads of the recommender: additive-attention pooling, user encoder + dot-product scorer, KD losses.
// All fp32 (raw exp without max-subtraction as in model_bert.py:27-32 needs fp32 range), reductions as
// wavefront reductions, fixed summation order (no float atomics).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// AttentionPooling over title tokens (model_bert.py:15-34, no mask).  One workgroup per title.
__global__ __launch_bounds__(256) void attpool_fwd_kernel(const bf16* __restrict__ y, const float* __restrict__ e,
                                                          int64_t lde, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, int Q, float* __restrict__ nv,
                                                          float* __restrict__ alpha, float* __restrict__ den, int L, int H) {
    __shared__ float al[512];
    const int Lr = (L + 31) & ~31;                 // alpha rows are padded to a multiple of 32 tokens
    const int64_t n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = w; i < L; i += 4) {
        const float* er = e + (n * L + i) * lde;
        float s = 0.f;
        for (int q = lane; q < Q; q += 64) s += er[q] * w2[q];
        s = wave_sum(s);
        if (lane == 0) al[i] = __expf(s + b2[0]);
    }
    __syncthreads();
    float d = 0.f;
    for (int i = 0; i < L; ++i) d += al[i];
    d += 1e-8f;
    for (int i = tid; i < Lr; i += 256) alpha[n * Lr + i] = i < L ? al[i] / d : 0.f;
    if (tid == 0) den[n] = d;
    for (int c = tid * 4; c < H; c += 1024) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < L; ++i) {
            bf16x4 v = *(const bf16x4*)(y + (n * L + i) * H + c);
            float wi = al[i] / d;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += wi * (float)v[r];
        }
        *(f32x4*)(nv + n * H + c) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
    }
}

__global__ __launch_bounds__(256) void attpool_bwd_kernel(const bf16* __restrict__ y, const float* __restrict__ e,
                                                          int64_t lde, const float* __restrict__ w2, int Q,
                                                          const float* __restrict__ dnv, const float* __restrict__ alpha,
                                                          bf16* __restrict__ dy, bf16* __restrict__ dpre, int64_t lddpre,
                                                          float* __restrict__ dw2_part, float* __restrict__ db2_part,
                                                          float* __restrict__ db1_part, int L, int H) {
    __shared__ float dw[512], da[512];
    __shared__ float Sred;
    const int Lr = (L + 31) & ~31;
    const int64_t n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = w; i < L; i += 4) {
        float s = 0.f;
        for (int c = lane * 4; c < H; c += 256) {
            bf16x4 v = *(const bf16x4*)(y + (n * L + i) * H + c);
            f32x4 g = *(const f32x4*)(dnv + n * H + c);
#pragma unroll
            for (int r = 0; r < 4; ++r) s += g[r] * (float)v[r];
        }
        s = wave_sum(s);
        if (lane == 0) dw[i] = s;
    }
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int i = 0; i < L; ++i) t += dw[i] * alpha[n * Lr + i];
        Sred = t;
    }
    __syncthreads();
    const float S = Sred;
    for (int i = tid; i < L; i += 256) da[i] = alpha[n * Lr + i] * (dw[i] - S);      // d loss / d (fc2 output) of token i
    __syncthreads();
    for (int c = tid * 4; c < H; c += 1024) {
        f32x4 g = *(const f32x4*)(dnv + n * H + c);
        for (int i = 0; i < L; ++i) {
            float wi = alpha[n * Lr + i];
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)(wi * g[r]);
            *(bf16x4*)(dy + (n * L + i) * H + c) = o;
        }
    }
    for (int q = tid; q < lddpre; q += 256) {
        float sw2 = 0.f, sb1 = 0.f;
        float wq = q < Q ? w2[q] : 0.f;
        for (int i = 0; i < L; ++i) {
            float ev = q < Q ? e[(n * L + i) * lde + q] : 0.f;
            bf16 dv = (bf16)(da[i] * wq * (1.f - ev * ev));
            dpre[(n * L + i) * lddpre + q] = dv;
            sb1 += (float)dv;
            sw2 += da[i] * ev;
        }
        if (q < Q) dw2_part[n * Q + q] = sw2;
        if (db1_part) db1_part[n * lddpre + q] = sb1;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < L; ++i) s += da[i];
        db2_part[n] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// The same pooling for FEW, LONG sequences (stage 1: 32 bodies of 128 ... 512 tokens).  One workgroup per sequence leaves 7 of 8
// CUs idle and walks the tokens in latency-bound loops (round-5 profile: 240 us backward / 131 us forward at 32 x 512); here every
// token-parallel part runs over (sequence, 64-token chunk) or one wave per token, and only the two sequence-wide scalars (the
// softmax-free denominator, the dot product S) are sums over the whole sequence - in token order, as above.  Chunk partials are
// combined in chunk order: deterministic, no atomics; the values differ from the one-workgroup kernels' by fp32 rounding only.
// ws (caller-owned, tnr_attpool_long_ws_elems floats): forward [n_seq][n_chunk][H] ; backward [n_seq][L] dw, then [n_seq][n_chunk][Q + lddpre + 1].
constexpr int AP_CH = 64;
__global__ __launch_bounds__(256) void attpool_long_score_kernel(const float* __restrict__ e, int64_t lde, const float* __restrict__ w2,
                                                                 const float* __restrict__ b2, int Q, float* __restrict__ alpha,
                                                                 int64_t n_tok, int L) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tok) return;
    const float* er = e + t * lde;
    float s = 0.f;
    for (int q = lane; q < Q; q += 64) s += er[q] * w2[q];
    s = wave_sum(s);
    const int Lr = (L + 31) & ~31;
    if (lane == 0) alpha[(t / L) * Lr + (t % L)] = __expf(s + b2[0]);          // unnormalised; attpool_long_fwd_fin divides
}
__global__ __launch_bounds__(256) void attpool_long_fwd_part_kernel(const bf16* __restrict__ y, const float* __restrict__ alpha,
                                                                    float* __restrict__ ws, int L, int H) {
    __shared__ float al[AP_CH];
    const int Lr = (L + 31) & ~31, nch = (L + AP_CH - 1) / AP_CH;
    const int64_t n = blockIdx.x;
    const int ch = blockIdx.y, i0 = ch * AP_CH, cnt = min(AP_CH, L - i0);
    if ((int)threadIdx.x < cnt) al[threadIdx.x] = alpha[n * Lr + i0 + threadIdx.x];
    __syncthreads();
    for (int c = threadIdx.x * 4; c < H; c += 1024) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < cnt; ++i) {
            bf16x4 v = *(const bf16x4*)(y + (n * L + i0 + i) * H + c);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += al[i] * (float)v[r];
        }
        *(f32x4*)(ws + (n * nch + ch) * H + c) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
    }
}
__global__ __launch_bounds__(256) void attpool_long_fwd_fin_kernel(const float* __restrict__ ws, float* __restrict__ alpha,
                                                                   float* __restrict__ nv, float* __restrict__ den, int L, int H) {
    __shared__ float dsh, al[512];
    const int Lr = (L + 31) & ~31, nch = (L + AP_CH - 1) / AP_CH;
    const int64_t n = blockIdx.x;
    for (int i = threadIdx.x; i < L; i += 256) al[i] = alpha[n * Lr + i];      // (512 dependent global reads took 25 us)
    __syncthreads();
    if (threadIdx.x == 0) {
        float d = 0.f;
        for (int i = 0; i < L; ++i) d += al[i];                                // the denominator, in token order
        dsh = d + 1e-8f;
        den[n] = d + 1e-8f;
    }
    __syncthreads();
    const float d = dsh;
    for (int c = threadIdx.x * 4; c < H; c += 1024) {
        f32x4 a = *(const f32x4*)(ws + (n * nch) * H + c);
        for (int ch = 1; ch < nch; ++ch) a += *(const f32x4*)(ws + (n * nch + ch) * H + c);
        *(f32x4*)(nv + n * H + c) = a / d;
    }
    for (int i = threadIdx.x; i < Lr; i += 256) alpha[n * Lr + i] = i < L ? al[i] / d : 0.f;
}
__global__ __launch_bounds__(256) void attpool_long_dw_kernel(const bf16* __restrict__ y, const float* __restrict__ dnv,
                                                              float* __restrict__ dwbuf, int64_t n_tok, int L, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tok) return;
    const int64_t n = t / L;
    float s = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        bf16x4 v = *(const bf16x4*)(y + t * H + c);
        f32x4 g = *(const f32x4*)(dnv + n * H + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) s += g[r] * (float)v[r];
    }
    s = wave_sum(s);
    if (lane == 0) dwbuf[t] = s;
}
__global__ __launch_bounds__(256) void attpool_long_bwd_part_kernel(const float* __restrict__ e, int64_t lde, const float* __restrict__ w2,
                                                                    int Q, const float* __restrict__ dnv, const float* __restrict__ alpha,
                                                                    const float* __restrict__ dwbuf, bf16* __restrict__ dy,
                                                                    bf16* __restrict__ dpre, int64_t lddpre, float* __restrict__ part,
                                                                    int L, int H) {
    __shared__ float prod[512], da[AP_CH], al[AP_CH];
    __shared__ float Sred;
    const int Lr = (L + 31) & ~31, nch = (L + AP_CH - 1) / AP_CH;
    const int64_t n = blockIdx.x;
    const int ch = blockIdx.y, i0 = ch * AP_CH, cnt = min(AP_CH, L - i0);
    const int tid = threadIdx.x;
    for (int i = tid; i < L; i += 256) prod[i] = dwbuf[n * L + i] * alpha[n * Lr + i];
    __syncthreads();
    if (tid == 0) {                                      // the sequence-wide dot product, in token order
        float t = 0.f;
        for (int i = 0; i < L; ++i) t += prod[i];
        Sred = t;
    }
    __syncthreads();
    if (tid < cnt) {
        const float a = alpha[n * Lr + i0 + tid];
        al[tid] = a;
        da[tid] = a * (dwbuf[n * L + i0 + tid] - Sred);
    }
    __syncthreads();
    for (int c = tid * 4; c < H; c += 1024) {
        f32x4 g = *(const f32x4*)(dnv + n * H + c);
        for (int i = 0; i < cnt; ++i) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)(al[i] * g[r]);
            *(bf16x4*)(dy + (n * L + i0 + i) * H + c) = o;
        }
    }
    float* const pr = part + (n * nch + ch) * (Q + lddpre + 1);
    for (int q = tid; q < lddpre; q += 256) {
        float sw2 = 0.f, sb1 = 0.f;
        const float wq = q < Q ? w2[q] : 0.f;
        for (int i = 0; i < cnt; ++i) {
            const float ev = q < Q ? e[(n * L + i0 + i) * lde + q] : 0.f;
            const bf16 dv = (bf16)(da[i] * wq * (1.f - ev * ev));
            dpre[(n * L + i0 + i) * lddpre + q] = dv;
            sb1 += (float)dv;
            sw2 += da[i] * ev;
        }
        if (q < Q) pr[q] = sw2;
        pr[Q + q] = sb1;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < cnt; ++i) s += da[i];
        pr[Q + lddpre] = s;
    }
}
__global__ __launch_bounds__(256) void attpool_long_bwd_fin_kernel(const float* __restrict__ part, int Q, int64_t lddpre, int nch,
                                                                   float* __restrict__ dw2_part, float* __restrict__ db2_part,
                                                                   float* __restrict__ db1_part) {
    const int64_t n = blockIdx.x;
    const int64_t W = Q + lddpre + 1;
    for (int64_t j = threadIdx.x; j < W; j += 256) {
        float s = part[(n * nch) * W + j];
        for (int ch = 1; ch < nch; ++ch) s += part[(n * nch + ch) * W + j];
        if (j < Q) dw2_part[n * Q + j] = s;
        else if (j < Q + lddpre) { if (db1_part) db1_part[n * lddpre + (j - Q)] = s; }
        else db2_part[n] = s;
    }
}

#ifndef TNR_BUILD_F16
// ------------------------------------------------------------------------------------------------
// small batched fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain): tile 64x64, BK 16
struct SgemmArgs {
    const float* A; int64_t a_rs, a_cs, sA;
    const float* B; int64_t b_rs, b_cs, sB;
    float* C; int64_t ldc, sC;
    const float* bias; int64_t sBias;
    int M, N, K;
    float alpha, beta;
    int ksplit, kchunk, batch;      // ksplit > 1: C = partials (ksplit, batch, M, N), no bias / beta
    float* A_out;                   // grouped launches: the destination of a split problem (its C is the partial buffer)
};

// One K step is 64 deep: 16 + 16 scalar loads per thread issued together (the next step's fly under this step's 32 MFMAs), so
// a step costs about one memory latency OR its MFMA time (32 x 64 cycles), whichever is longer -- with 16-deep steps every
// step paid a full latency for 8 MFMAs (44 us for the 1760 x 256 x 768 dense layer, 13 us for an M = 1 call).
constexpr int SG_BK = 64;
__device__ __forceinline__ void sgemm_tile(const SgemmArgs& g, int bx, int by, int bz, float (&As)[64][SG_BK + 1], float (&Bs)[64][SG_BK + 1]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int zb = bz / g.ksplit, zk = bz - zb * g.ksplit;
    const float* A = g.A + zb * g.sA;
    const float* B = g.B + zb * g.sB;
    float* C = g.ksplit > 1 ? g.C + ((int64_t)zk * g.batch + zb) * g.M * g.N : g.C + zb * g.sC;
    const int64_t ldc = g.ksplit > 1 ? g.N : g.ldc;
    const int m0 = by * 64, n0 = bx * 64;
    const int kbeg = zk * g.kchunk;
    const int kend = kbeg + g.kchunk < g.K ? kbeg + g.kchunk : g.K;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool a_kfast = g.a_cs == 1, b_kfast = g.b_cs == 1;
    constexpr int NQ = 64 * SG_BK / 256;         // elements per thread, operand and step
    // k-fast operands: 64 consecutive threads read one row's 64 k values; otherwise 64 consecutive threads read 64 rows of one k
    const int ra0 = a_kfast ? tid >> 6 : tid & 63, ka0 = a_kfast ? tid & 63 : tid >> 6;
    const int rb0 = b_kfast ? tid >> 6 : tid & 63, kb0 = b_kfast ? tid & 63 : tid >> 6;
    float av[NQ], bv[NQ];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ra = a_kfast ? ra0 + 4 * q : ra0, ka = a_kfast ? ka0 : ka0 + 4 * q;
            const int rb = b_kfast ? rb0 + 4 * q : rb0, kb = b_kfast ? kb0 : kb0 + 4 * q;
            av[q] = (m0 + ra < g.M && k0 + ka < kend) ? A[(int64_t)(m0 + ra) * g.a_rs + (int64_t)(k0 + ka) * g.a_cs] : 0.f;
            bv[q] = (n0 + rb < g.N && k0 + kb < kend) ? B[(int64_t)(n0 + rb) * g.b_rs + (int64_t)(k0 + kb) * g.b_cs] : 0.f;
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += SG_BK) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ra = a_kfast ? ra0 + 4 * q : ra0, ka = a_kfast ? ka0 : ka0 + 4 * q;
            const int rb = b_kfast ? rb0 + 4 * q : rb0, kb = b_kfast ? kb0 : kb0 + 4 * q;
            As[ra][ka] = av[q];
            Bs[rb][kb] = bv[q];
        }
        __syncthreads();
        if (k0 + SG_BK < kend) fetch(k0 + SG_BK);    // next step's global loads fly under the MFMAs
        const int kk_end = kend - k0 < SG_BK ? kend - k0 : SG_BK;
        for (int kk = 0; kk < kk_end; kk += 2) {      // fixed order: an fp32 fma chain over k (zero padding beyond kend)
            float a = As[wm * 32 + (lane & 31)][kk + (lane >> 5)];
            float b = Bs[wn * 32 + (lane & 31)][kk + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int n = n0 + wn * 32 + (lane & 31);
    if (n >= g.N) return;
    const float bias = (g.bias && g.ksplit == 1) ? g.bias[zb * g.sBias + n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M) {
            float* c = C + (int64_t)m * ldc + n;
            if (g.ksplit > 1) {
                *c = acc[r];
            } else {
                float v = g.alpha * acc[r] + bias;
                if (g.beta != 0.f) v += g.beta * *c;
                *c = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void sgemm_kernel(SgemmArgs g) {
    __shared__ float As[64][SG_BK + 1], Bs[64][SG_BK + 1];
    sgemm_tile(g, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Several independent GEMMs in ONE launch (tnr_sgemm_group): the heads' fp32 GEMMs are latency-sized (a workgroup's time is its
// K loop, few tiles each), so problems that do not depend on each other cost the longest of them instead of their sum.
constexpr int SG_MAXP = 8;
struct SgemmGroup {
    SgemmArgs a[SG_MAXP];
    int start[SG_MAXP + 1];        // first block of each problem ; start[n] = grid size
    int gx[SG_MAXP], gy[SG_MAXP];  // column / row tiles of each problem
    int64_t rstart[SG_MAXP + 1];   // split-K reduce: first element of each problem (problems without a split have none)
    float alpha[SG_MAXP], beta[SG_MAXP];
    int n;
};
__global__ __launch_bounds__(256) void sgemm_group_kernel(SgemmGroup g) {
    __shared__ float As[64][SG_BK + 1], Bs[64][SG_BK + 1];
    const int b = blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < SG_MAXP; ++k)
        if (k < g.n && b >= g.start[k]) i = k;
    const int l = b - g.start[i];
    const int per = g.gx[i] * g.gy[i];
    const int bz = l / per, r = l - bz * per;
    const int by = r / g.gx[i], bx = r - by * g.gx[i];
    sgemm_tile(g.a[i], bx, by, bz, As, Bs);
}
__global__ __launch_bounds__(256) void sgemm_group_reduce_kernel(SgemmGroup g) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.rstart[g.n]) return;
    int i = 0;
#pragma unroll
    for (int k = 1; k < SG_MAXP; ++k)
        if (k < g.n && e >= g.rstart[k]) i = k;
    const SgemmArgs& a = g.a[i];
    const int64_t per = (int64_t)a.M * a.N, total = per * a.batch, j = e - g.rstart[i];
    const float* part = a.C;                             // split problems carry their partial buffer in C ...
    float t = part[j];
    for (int s = 1; s < a.ksplit; ++s) t += part[(int64_t)s * total + j];
    const int z = (int)(j / per);
    const int64_t rr = j - (int64_t)z * per;
    const int m = (int)(rr / a.N), n = (int)(rr - (int64_t)m * a.N);
    float* c = (float*)a.A_out + (int64_t)z * a.sC + (int64_t)m * a.ldc + n;      // ... and the real output here
    float v = g.alpha[i] * t + (a.bias ? a.bias[(int64_t)z * a.sBias + n] : 0.f);
    if (g.beta[i] != 0.f) v += g.beta[i] * *c;
    *c = v;
}

// split-K epilogue: C_z[m,n] = alpha * sum_s part[s][z][m][n] + bias_z[n] + beta * C_z[m,n], slices summed in order
__global__ __launch_bounds__(256) void sgemm_reduce_kernel(const float* __restrict__ part, int ksplit, int batch, int M, int N,
                                                           float* __restrict__ C, int64_t ldc, int64_t sC,
                                                           const float* __restrict__ bias, int64_t sBias, float alpha, float beta) {
    const int64_t per = (int64_t)M * N, total = per * batch;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float t = part[i];
    for (int s = 1; s < ksplit; ++s) t += part[(int64_t)s * total + i];
    const int z = (int)(i / per);
    const int64_t r = i - (int64_t)z * per;
    const int m = (int)(r / N), n = (int)(r - (int64_t)m * N);
    float* c = C + (int64_t)z * sC + (int64_t)m * ldc + n;
    float v = alpha * t + (bias ? bias[(int64_t)z * sBias + n] : 0.f);
    if (beta != 0.f) v += beta * *c;
    *c = v;
}

// rows of per-model tables gathered into a dense (n_model, n_idx, D) array
__global__ void gather_rows_kernel(const float* __restrict__ tbl, int64_t R, const int32_t* __restrict__ idx,
                                   int64_t n_idx, int D, float* __restrict__ out, int64_t out_rows, int64_t out_row0) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int d4 = D / 4;
    int64_t total = n_idx * d4;
    if (t >= total) return;
    int z = blockIdx.y;
    int64_t r = t / d4;
    int c = (int)(t - r * d4) * 4;
    *(f32x4*)(out + ((int64_t)z * out_rows + out_row0 + r) * D + c) =
        *(const f32x4*)(tbl + ((int64_t)z * R + idx[r]) * D + c);
}


// out[u, :] = sum over j in [seg[u], seg[u+1]) of src[order[j], :], summed in a fixed order (in-batch de-duplication:
// the gradient of a news vector that several history / candidate slots share).  One 1024-thread workgroup per output row:
// wave w takes entries w, w+16, ... with four independent accumulators, the 16 partial rows are added in wave order.
__global__ __launch_bounds__(1024) void segment_sum_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ order,
                                                                const int32_t* __restrict__ seg, int D, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [16][D]
    const int u = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j0 = seg[u], j1 = seg[u + 1];
    if (j1 - j0 <= 4) {                                              // common case: no exchange needed
        if (w == 0)
            for (int c = lane * 4; c < D; c += 256) {
                f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int j = j0; j < j1; ++j) a += *(const f32x4*)(src + (int64_t)order[j] * D + c);
                *(f32x4*)(out + (int64_t)u * D + c) = a;
            }
        return;
    }
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int j = j0 + w;
        for (; j + 48 < j1; j += 64) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += *(const f32x4*)(src + (int64_t)order[j + 16 * q] * D + c);
        }
        for (int q = 0; j < j1; j += 16, ++q) a[q & 3] += *(const f32x4*)(src + (int64_t)order[j] * D + c);
        *(f32x4*)(sm + w * D + c) = (a[0] + a[1]) + (a[2] + a[3]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 1024) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sm[q * D + c];
        out[(int64_t)u * D + c] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// UserEncoder (model_bert.py:155-176) + scorer (:204).  One workgroup per (impression, model).
constexpr int MAXU = 64;

template <bool FWD>
__device__ __forceinline__ void load_hv(float* hv, const float* vec, const int32_t* hidx, const float* mask,
                                        const float* pad, int user_log_mask, int U, int D, int tid) {
    for (int t = tid; t < U * (D / 4); t += 256) {
        int u = t / (D / 4), c = (t - u * (D / 4)) * 4;
        f32x4 v = *(const f32x4*)(vec + (int64_t)hidx[u] * D + c);
        if (!user_log_mask) {
            float m = mask[u];
            f32x4 p = *(const f32x4*)(pad + c);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] * m + p[r] * (1.f - m);
        }
        *(f32x4*)(hv + u * D + c) = v;
    }
}

// fc1 of the additive attention is done beforehand as one batched GEMM over all (impression, slot) rows
// (epre = v W1^T + b1, position order b*U+u); slots replaced by pad_doc (user_log_mask False) take fc1(pad).
__global__ __launch_bounds__(256) void user_score_fwd_kernel(
    const float* __restrict__ vec, int64_t R, const int32_t* __restrict__ hidx, const int32_t* __restrict__ cidx,
    const float* __restrict__ mask, const float* __restrict__ pad, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, int user_log_mask,
    const float* __restrict__ epre, const float* __restrict__ epad_in, float* __restrict__ user, int64_t user_stride,
    float* __restrict__ score, float* __restrict__ e_out, float* __restrict__ alpha, float* __restrict__ den, int B, int U,
    int C, int D, int Q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hv = (float*)smem;                 // [U][D]
    float* es = hv + U * D;                   // [U][Q]
    float* al = es + U * Q;                   // [MAXU]
    float* us = al + MAXU;                    // [D]
    float* epad = us + D;                     // [Q]
    int& any_pad = *(int*)(epad + Q);         // all LDS in the dynamic region (static LDS would eat into the 160 KB cap)
    const int b = blockIdx.x, z = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    vec += (int64_t)z * R * D;
    pad += (int64_t)z * D;
    w1 += (int64_t)z * Q * D;
    b1 += (int64_t)z * Q;
    w2 += (int64_t)z * Q;
    hidx += (int64_t)b * U;
    cidx += (int64_t)b * C;
    mask += (int64_t)b * U;
    epre += ((int64_t)z * B + b) * U * Q;
    if (tid == 0) {
        int a = 0;
        if (!user_log_mask)
            for (int u = 0; u < U; ++u) a |= (mask[u] == 0.f);
        any_pad = a;
    }
    load_hv<true>(hv, vec, hidx, mask, pad, user_log_mask, U, D, tid);
    __syncthreads();
    if (any_pad) {
        // fc1(pad_doc) depends on the model only: taken from the caller (one M=1 tnr_sgemm per step) when given
        if (epad_in) {
            for (int q = tid; q < Q; q += 256) epad[q] = epad_in[(int64_t)z * Q + q];
        } else {
            for (int q = tid; q < Q; q += 256) {
                const float* wr = w1 + (int64_t)q * D;
                float s = 0.f;
                for (int d = 0; d < D; d += 4) {
                    f32x4 wv = *(const f32x4*)(wr + d), pv = *(const f32x4*)(pad + d);
                    s += wv[0] * pv[0] + wv[1] * pv[1] + wv[2] * pv[2] + wv[3] * pv[3];
                }
                epad[q] = s + b1[q];
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < U * Q; idx += 256) {
        int u = idx / Q, q = idx - u * Q;
        float pre = (user_log_mask || mask[u] != 0.f) ? epre[idx] : epad[q];
        float ev = tnr_tanh(pre);
        es[idx] = ev;
        e_out[(((int64_t)z * B + b) * U) * Q + idx] = ev;
    }
    __syncthreads();
    for (int u = w; u < U; u += 4) {
        float s = 0.f;
        for (int q = lane; q < Q; q += 64) s += es[u * Q + q] * w2[q];
        s = wave_sum(s);
        if (lane == 0) {
            float a = __expf(s + b2[z]);
            if (user_log_mask) a *= mask[u];
            al[u] = a;
        }
    }
    __syncthreads();
    float dsum = 0.f;
    for (int u = 0; u < U; ++u) dsum += al[u];
    dsum += 1e-8f;
    if (tid < U) alpha[((int64_t)z * B + b) * U + tid] = al[tid] / dsum;
    if (tid == 0) den[(int64_t)z * B + b] = dsum;
    for (int d = tid; d < D; d += 256) {
        float s = 0.f;
        for (int u = 0; u < U; ++u) s += (al[u] / dsum) * hv[u * D + d];
        us[d] = s;
        user[(int64_t)z * user_stride + (int64_t)b * D + d] = s;
    }
    __syncthreads();
    for (int c = w; c < C; c += 4) {
        const float* cr = vec + (int64_t)cidx[c] * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += cr[d] * us[d];
        s = wave_sum(s);
        if (lane == 0) score[((int64_t)z * B + b) * C + c] = s;
    }
}

// The same forward with fc1 INSIDE (epre == NULL at the entry point): one launch per user-encoder pass instead of a batched fp32
// GEMM + its split-K reduce + an M = 1 GEMM pair for fc1(pad_doc) + the kernel above (77 us of launches for 0.17 GFLOP at
// B = 32).  One workgroup per (impression, model) as before; the blended history rows sit in LDS ([64][D + 4], rows >= U zero)
// and are the A operand of 32x32x2 fp32 MFMAs against W1 rows read straight from global memory (205 KB per model, L2-resident
// across the impressions): lane (j, h) of a 32 x 32 (slot, unit) block takes k = 8 t + 4 h .. + 3 of both operands with one
// 16-byte access each and feeds four MFMAs - a fixed permutation of the fp32 sum over k.  fc1 of a masked slot is fc1 of the
// blended row (= fc1(pad_doc) for a 0 / 1 mask, model_bert.py:162-164), so there is no separate pad path.
constexpr int UF_XTRA = 4;                     // LDS row pitch D + 4 floats: 16-byte aligned rows, banks shifted by 4 per row
constexpr int UF_THREADS = 1024;               // four waves per SIMD: the 14 (slot block, unit block) pairs of U = 50, Q = 200 in ONE round of
                                               // 16 waves (512 threads: two rounds of 8; 42 -> 37 us), a wave's W1 loads under the others' MFMAs
__global__ __launch_bounds__(UF_THREADS) void user_fwd_fused_kernel(
    const float* __restrict__ vec, int64_t R, const int32_t* __restrict__ hidx, const int32_t* __restrict__ cidx,
    const float* __restrict__ mask, const float* __restrict__ pad, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, int user_log_mask,
    float* __restrict__ user, int64_t user_stride, float* __restrict__ score, float* __restrict__ e_out,
    float* __restrict__ alpha, float* __restrict__ den, int B, int U, int C, int D, int Q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int P = D + UF_XTRA;
    float* hv = (float*)smem;                 // [64][P]
    float* es = hv + 64 * P;                  // [U][Q]
    float* al = es + U * Q;                   // [MAXU]
    float* us = al + MAXU;                    // [D]
    const int b = blockIdx.x, z = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    vec += (int64_t)z * R * D;
    pad += (int64_t)z * D;
    w1 += (int64_t)z * Q * D;
    b1 += (int64_t)z * Q;
    w2 += (int64_t)z * Q;
    hidx += (int64_t)b * U;
    cidx += (int64_t)b * C;
    mask += (int64_t)b * U;
    for (int t = tid; t < 64 * (D / 4); t += UF_THREADS) {
        const int u = t / (D / 4), c = (t - u * (D / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (u < U) {
            v = *(const f32x4*)(vec + (int64_t)hidx[u] * D + c);
            if (!user_log_mask) {
                const float m = mask[u];
                const f32x4 p = *(const f32x4*)(pad + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * m + p[r] * (1.f - m);
            }
        }
        *(f32x4*)(hv + u * P + c) = v;
    }
    __syncthreads();
    // e = tanh(hv W1^T + b1): (slot block, unit block) pairs round-robin over the waves
    const int nub = (U + 31) >> 5, nqb = (Q + 31) >> 5;
    const int j = lane & 31, h = lane >> 5;
    for (int pr = w; pr < nub * nqb; pr += UF_THREADS / 64) {
        const int ub = pr % nub, qb = pr / nub;
        const int q = qb * 32 + j;
        const float* ap = hv + (ub * 32 + j) * P + 4 * h;
        const float* bp = w1 + (int64_t)(q < Q ? q : Q - 1) * D + 4 * h;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int t = 0;
        f32x4 bn[4];                               // the W1 operand of the NEXT 32 k values: requested before this group's MFMAs
        if (D >= 32) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bn[g] = *(const f32x4*)(bp + 8 * g);
        }
        for (; t + 32 <= D; t += 32) {
            f32x4 a4[4], b4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                a4[g] = *(const f32x4*)(ap + t + 8 * g);
                b4[g] = bn[g];
            }
            if (t + 64 <= D) {
#pragma unroll
                for (int g = 0; g < 4; ++g) bn[g] = *(const f32x4*)(bp + t + 32 + 8 * g);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g][m], b4[g][m], acc, 0, 0, 0);
        }
        for (; t < D; t += 8) {
            const f32x4 a4 = *(const f32x4*)(ap + t);
            const f32x4 b4 = *(const f32x4*)(bp + t);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m], b4[m], acc, 0, 0, 0);
        }
        if (q < Q) {
            const float bq = b1[q];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = ub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (u < U) {
                    const float ev = tnr_tanh(acc[r] + bq);
                    es[u * Q + q] = ev;
                    e_out[(((int64_t)z * B + b) * U + u) * Q + q] = ev;
                }
            }
        }
    }
    __syncthreads();
    for (int u = w; u < U; u += UF_THREADS / 64) {
        float s = 0.f;
        for (int q = lane; q < Q; q += 64) s += es[u * Q + q] * w2[q];
        s = wave_sum(s);
        if (lane == 0) {
            float a = __expf(s + b2[z]);
            if (user_log_mask) a *= mask[u];
            al[u] = a;
        }
    }
    __syncthreads();
    float dsum = 0.f;
    for (int u = 0; u < U; ++u) dsum += al[u];
    dsum += 1e-8f;
    __syncthreads();                           // everyone has read the un-normalised weights
    if (tid < U) {
        const float a = al[tid] / dsum;
        al[tid] = a;
        alpha[((int64_t)z * B + b) * U + tid] = a;
    }
    if (tid == 0) den[(int64_t)z * B + b] = dsum;
    __syncthreads();
    for (int d = tid; d < D; d += UF_THREADS) {
        float s = 0.f;
        for (int u = 0; u < U; ++u) s += al[u] * hv[u * P + d];
        us[d] = s;
        user[(int64_t)z * user_stride + (int64_t)b * D + d] = s;
    }
    __syncthreads();
    for (int c = w; c < C; c += UF_THREADS / 64) {
        const float* cr = vec + (int64_t)cidx[c] * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += cr[d] * us[d];
        s = wave_sum(s);
        if (lane == 0) score[((int64_t)z * B + b) * C + c] = s;
    }
}

// backward of the student's user encoder, split so that its two contractions (0.33 GFLOP at B=32: the fc1 weight
// gradient and the gradient w.r.t. the blended history vectors) run on the fp32 MFMA GEMM instead of per-thread FMA
// chains (the single-kernel version spent 156 us on 32 workgroups):
//   pre : hv (blended history rows, position order) ; dpre = d tanh-preactivation (B*U, Q) ; per-impression
//         partials [b1 | w2 | pad | b2] (pad filled by post)
//   host: dW1 = dpre^T hv (tnr_sgemm, split-K) ; dhv = dpre W1 (tnr_sgemm)
//   post: dvec[hidx] += (alpha * duser + dhv) * m ; pad partial = sum_u (...) * (1 - m)
__global__ __launch_bounds__(256) void user_bwd_pre_kernel(
    const float* __restrict__ vec, const int32_t* __restrict__ hidx, const float* __restrict__ mask,
    const float* __restrict__ pad, const float* __restrict__ w2, int user_log_mask, const float* __restrict__ duser,
    const float* __restrict__ e, const float* __restrict__ alpha, float* __restrict__ hv_out, float* __restrict__ dpre,
    float* __restrict__ part, int U, int D, int Q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hv = (float*)smem;                 // [U][D]
    float* dw = hv + U * D;                   // [MAXU]
    float* da = dw + MAXU;                    // [MAXU]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    hidx += (int64_t)b * U;
    mask += (int64_t)b * U;
    duser += (int64_t)b * D;
    e += (int64_t)b * U * Q;
    alpha += (int64_t)b * U;
    dpre += (int64_t)b * U * Q;
    hv_out += (int64_t)b * U * D;
    float* p_b1 = part + (int64_t)b * (2 * Q + D + 1);
    float* p_w2 = p_b1 + Q;
    float* p_b2 = p_w2 + Q + D;
    load_hv<false>(hv, vec, hidx, mask, pad, user_log_mask, U, D, tid);
    __syncthreads();
    for (int t = tid * 4; t < U * D; t += 1024) *(f32x4*)(hv_out + t) = *(const f32x4*)(hv + t);
    for (int u = w; u < U; u += 4) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += duser[d] * hv[u * D + d];
        s = wave_sum(s);
        if (lane == 0) dw[u] = s;
    }
    __syncthreads();
    float S = 0.f;
    for (int u = 0; u < U; ++u) S += dw[u] * alpha[u];
    if (tid < MAXU) da[tid] = tid < U ? alpha[tid] * (dw[tid] - S) : 0.f;
    __syncthreads();
    for (int q = tid; q < Q; q += 256) {
        float wq = w2[q], sb1 = 0.f, sw2 = 0.f;
        for (int u = 0; u < U; ++u) {
            float ev = e[u * Q + q];
            float v = da[u] * wq * (1.f - ev * ev);
            sw2 += da[u] * ev;
            dpre[u * Q + q] = v;
            sb1 += v;
        }
        p_b1[q] = sb1;
        p_w2[q] = sw2;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int u = 0; u < U; ++u) s += da[u];
        p_b2[0] = s;
    }
}

__global__ __launch_bounds__(256) void user_bwd_post_kernel(
    const float* __restrict__ dhv, const float* __restrict__ alpha, const float* __restrict__ duser,
    const float* __restrict__ mask, const int32_t* __restrict__ hidx, int user_log_mask, float* __restrict__ dvec,
    float* __restrict__ part, int U, int D, int Q) {
    const int b = blockIdx.x;
    hidx += (int64_t)b * U;
    mask += (int64_t)b * U;
    alpha += (int64_t)b * U;
    dhv += (int64_t)b * U * D;
    float* p_pad = part + (int64_t)b * (2 * Q + D + 1) + 2 * Q;
    for (int d = threadIdx.x; d < D; d += 256) {
        float du = duser[(int64_t)b * D + d], dp_pad = 0.f;
        for (int u = 0; u < U; ++u) {
            float g = alpha[u] * du + dhv[u * D + d];
            float m = user_log_mask ? 1.f : mask[u];
            dp_pad += g * (1.f - m);
            dvec[(int64_t)hidx[u] * D + d] += g * m;
        }
        p_pad[d] = dp_pad;
    }
}

// ------------------------------------------------------------------------------------------------
// NRMS user encoder (args.model == 'NRMS', model_bert.py:37-100, 145-148, 162-164, 171-173): a multi-head
// self-attention (d_k = d_v = 16) over the U clicked-news vectors in front of the additive pooling.
//   blend : hv[z, b*U+u] = vec[z, hidx[b,u]] * m + pad[z] * (1 - m)   (user_log_mask False) | vec row (True)
//   qkv   = hv [W_Q; W_K; W_V]^T + b     (tnr_sgemm, batched over the models)
//   attn  : ctx_i = sum_j sc_ij v_j / (sum_j sc_ij + 1e-8), sc_ij = exp(q_i.k_j / 4) [* m_j]   -- raw exp (:51-58)
// Backward (student only) recomputes sc from q, k in two deterministic phases (rows, then columns; no atomics).
constexpr int NRMS_DK = 16;

__global__ __launch_bounds__(256) void user_blend_fwd_kernel(const float* __restrict__ vec, int64_t R,
                                                             const int32_t* __restrict__ hidx, const float* __restrict__ mask,
                                                             const float* __restrict__ pad, int user_log_mask,
                                                             float* __restrict__ hv, int B, int U, int D) {
    const int z = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int d4 = D / 4;
    if (t >= (int64_t)B * U * d4) return;
    const int64_t r = t / d4;
    const int c = (int)(t - r * d4) * 4;
    f32x4 v = *(const f32x4*)(vec + ((int64_t)z * R + hidx[r]) * D + c);
    if (!user_log_mask) {
        float m = mask[r];
        f32x4 p = *(const f32x4*)(pad + (int64_t)z * D + c);
        v = v * m + p * (1.f - m);
    }
    *(f32x4*)(hv + ((int64_t)z * B * U + r) * D + c) = v;
}

// dvec[hidx[b,u]] += dhv * m ; pad partial (per impression) = sum_u dhv * (1 - m)        (user_log_mask False)
// dvec[hidx[b,u]] += dhv                                                                   (user_log_mask True)
__global__ __launch_bounds__(256) void user_blend_bwd_kernel(const float* __restrict__ dhv, const float* __restrict__ mask,
                                                             const int32_t* __restrict__ hidx, int user_log_mask,
                                                             float* __restrict__ dvec, float* __restrict__ pad_part,
                                                             int64_t part_stride, int U, int D) {
    const int b = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) {
        float dp = 0.f;
        for (int u = 0; u < U; ++u) {
            float g = dhv[((int64_t)b * U + u) * D + d];
            float m = user_log_mask ? 1.f : mask[(int64_t)b * U + u];
            dp += g * (1.f - m);
            dvec[(int64_t)hidx[(int64_t)b * U + u] * D + d] += g * m;
        }
        pad_part[(int64_t)b * part_stride + d] = dp;
    }
}

// one workgroup per (impression, model): K and V of every head in LDS, one thread per (query i, head h)
__global__ __launch_bounds__(256) void nrms_attn_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ mask,
                                                            int use_mask, float* __restrict__ ctx, int64_t ctx_rows, int B,
                                                            int U, int NH) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Dh = NH * NRMS_DK;
    float* Ks = sm;                     // [U][Dh]
    float* Vs = sm + U * Dh;            // [U][Dh]
    const int b = blockIdx.x, z = blockIdx.y;
    const float* base = qkv + ((int64_t)z * B + b) * U * 3 * Dh;
    for (int t = threadIdx.x * 4; t < U * Dh; t += 1024) {
        int u = t / Dh, c = t - u * Dh;
        *(f32x4*)(Ks + t) = *(const f32x4*)(base + (int64_t)u * 3 * Dh + Dh + c);
        *(f32x4*)(Vs + t) = *(const f32x4*)(base + (int64_t)u * 3 * Dh + 2 * Dh + c);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < U * NH; p += 256) {
        const int i = p / NH, h = p - i * NH;
        float q[NRMS_DK], acc[NRMS_DK];
#pragma unroll
        for (int d = 0; d < NRMS_DK; ++d) {
            q[d] = base[(int64_t)i * 3 * Dh + h * NRMS_DK + d] * 0.25f;       // 1 / sqrt(16)
            acc[d] = 0.f;
        }
        float den = 0.f;
        for (int j = 0; j < U; ++j) {
            const float* kj = Ks + j * Dh + h * NRMS_DK;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < NRMS_DK; ++d) s += q[d] * kj[d];
            float e = __expf(s);
            if (use_mask) e *= mask[(int64_t)b * U + j];
            den += e;
            const float* vj = Vs + j * Dh + h * NRMS_DK;
#pragma unroll
            for (int d = 0; d < NRMS_DK; ++d) acc[d] += e * vj[d];
        }
        const float inv = 1.f / (den + 1e-8f);
        float* o = ctx + ((int64_t)z * ctx_rows + (int64_t)b * U + i) * Dh + h * NRMS_DK;
#pragma unroll
        for (int d = 0; d < NRMS_DK; ++d) o[d] = acc[d] * inv;
    }
}

__global__ __launch_bounds__(256) void nrms_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ mask,
                                                            int use_mask, const float* __restrict__ dctx,
                                                            float* __restrict__ dqkv, int U, int NH) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Dh = NH * NRMS_DK;
    float* A0 = sm;                     // phase A: K        phase B: Q * 0.25
    float* A1 = sm + U * Dh;            // phase A: V        phase B: dctx
    float* dens = A1 + U * Dh;          // [U][NH]  sum_j sc_ij + 1e-8
    float* rs = dens + U * NH;          // [U][NH]  sum_j attn_ij * (dctx_i . v_j)
    const int b = blockIdx.x;
    const float* base = qkv + (int64_t)b * U * 3 * Dh;
    const float* dcb = dctx + (int64_t)b * U * Dh;
    float* dbase = dqkv + (int64_t)b * U * 3 * Dh;
    for (int t = threadIdx.x * 4; t < U * Dh; t += 1024) {
        int u = t / Dh, c = t - u * Dh;
        *(f32x4*)(A0 + t) = *(const f32x4*)(base + (int64_t)u * 3 * Dh + Dh + c);
        *(f32x4*)(A1 + t) = *(const f32x4*)(base + (int64_t)u * 3 * Dh + 2 * Dh + c);
    }
    __syncthreads();
    // phase A: per query row -- den_i, r_i, dq_i
    for (int p = threadIdx.x; p < U * NH; p += 256) {
        const int i = p / NH, h = p - i * NH;
        float q[NRMS_DK], dc[NRMS_DK], dq[NRMS_DK];
#pragma unroll
        for (int d = 0; d < NRMS_DK; ++d) {
            q[d] = base[(int64_t)i * 3 * Dh + h * NRMS_DK + d] * 0.25f;
            dc[d] = dcb[(int64_t)i * Dh + h * NRMS_DK + d];
            dq[d] = 0.f;
        }
        float den = 0.f, t1 = 0.f;
        for (int j = 0; j < U; ++j) {
            const float* kj = A0 + j * Dh + h * NRMS_DK;
            const float* vj = A1 + j * Dh + h * NRMS_DK;
            float s = 0.f, da = 0.f;
#pragma unroll
            for (int d = 0; d < NRMS_DK; ++d) {
                s += q[d] * kj[d];
                da += dc[d] * vj[d];
            }
            float e = __expf(s);
            if (use_mask) e *= mask[(int64_t)b * U + j];
            den += e;
            t1 += e * da;
        }
        den += 1e-8f;
        const float r = t1 / den;
        dens[p] = den;
        rs[p] = r;
        for (int j = 0; j < U; ++j) {
            const float* kj = A0 + j * Dh + h * NRMS_DK;
            const float* vj = A1 + j * Dh + h * NRMS_DK;
            float s = 0.f, da = 0.f;
#pragma unroll
            for (int d = 0; d < NRMS_DK; ++d) {
                s += q[d] * kj[d];
                da += dc[d] * vj[d];
            }
            float e = __expf(s);
            if (use_mask) e *= mask[(int64_t)b * U + j];
            const float ds = (da - r) / den * e * 0.25f;
#pragma unroll
            for (int d = 0; d < NRMS_DK; ++d) dq[d] += ds * kj[d];
        }
        float* o = dbase + (int64_t)i * 3 * Dh + h * NRMS_DK;
#pragma unroll
        for (int d = 0; d < NRMS_DK; ++d) o[d] = dq[d];
    }
    __syncthreads();
    // phase B: per key column -- dk_j, dv_j
    for (int t = threadIdx.x * 4; t < U * Dh; t += 1024) {
        int u = t / Dh, c = t - u * Dh;
        f32x4 qv = *(const f32x4*)(base + (int64_t)u * 3 * Dh + c);
        *(f32x4*)(A0 + t) = qv * 0.25f;
        *(f32x4*)(A1 + t) = *(const f32x4*)(dcb + (int64_t)u * Dh + c);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < U * NH; p += 256) {
        const int j = p / NH, h = p - j * NH;
        float k[NRMS_DK], v[NRMS_DK], dk[NRMS_DK], dv[NRMS_DK];
#pragma unroll
        for (int d = 0; d < NRMS_DK; ++d) {
            k[d] = base[(int64_t)j * 3 * Dh + Dh + h * NRMS_DK + d];
            v[d] = base[(int64_t)j * 3 * Dh + 2 * Dh + h * NRMS_DK + d];
            dk[d] = 0.f;
            dv[d] = 0.f;
        }
        const float mj = use_mask ? mask[(int64_t)b * U + j] : 1.f;
        for (int i = 0; i < U; ++i) {
            const float* qi = A0 + i * Dh + h * NRMS_DK;
            const float* di = A1 + i * Dh + h * NRMS_DK;
            float s = 0.f, da = 0.f;
#pragma unroll
            for (int d = 0; d < NRMS_DK; ++d) {
                s += qi[d] * k[d];
                da += di[d] * v[d];
            }
            const float e = __expf(s) * mj;
            const float den = dens[i * NH + h];
            const float a = e / den;
            const float ds = (da - rs[i * NH + h]) / den * e;       // d score (before the 1/4): qi already carries 1/4
#pragma unroll
            for (int d = 0; d < NRMS_DK; ++d) {
                dk[d] += ds * qi[d];
                dv[d] += a * di[d];
            }
        }
        float* ok = dbase + (int64_t)j * 3 * Dh + Dh + h * NRMS_DK;
        float* ov = dbase + (int64_t)j * 3 * Dh + 2 * Dh + h * NRMS_DK;
#pragma unroll
        for (int d = 0; d < NRMS_DK; ++d) {
            ok[d] = dk[d];
            ov[d] = dv[d];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// score-level KD losses (model_bert.py:271, 286-298): one thread per impression, fixed-order block sum
__global__ __launch_bounds__(256) void kd_score_loss_kernel(const float* __restrict__ s_score,
                                                            const float* __restrict__ t_score,
                                                            const int64_t* __restrict__ label, float tau, float coef,
                                                            float* __restrict__ tw, float* __restrict__ dscore,
                                                            float* __restrict__ losses, int B, int C, int T) {
    __shared__ float red[2][256];
    float l_distill = 0.f, l_target = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const int y = (int)label[b];
        // teacher CE and weights softmax(-CE)
        float ce[16], mxw = -3.0e38f;
        for (int i = 0; i < T; ++i) {
            const float* ts = t_score + ((int64_t)i * B + b) * C;
            float mx = -3.0e38f;
            for (int c = 0; c < C; ++c) mx = fmaxf(mx, ts[c]);
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += __expf(ts[c] - mx);
            ce[i] = -(ts[y] - mx - __logf(se));
            mxw = fmaxf(mxw, -ce[i]);
        }
        float sw = 0.f;
        for (int i = 0; i < T; ++i) {
            ce[i] = __expf(-ce[i] - mxw);
            sw += ce[i];
        }
        for (int i = 0; i < T; ++i) {
            ce[i] /= sw;
            tw[(int64_t)b * T + i] = ce[i];
        }
        // mixed teacher scores, student log-softmax at temperature tau and at 1
        const float* ss = s_score + (int64_t)b * C;
        float mt = -3.0e38f, ms = -3.0e38f, ms1 = -3.0e38f;
        float tm[16];
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
            for (int i = 0; i < T; ++i) v += t_score[((int64_t)i * B + b) * C + c] * ce[i];
            tm[c] = v / tau;
            mt = fmaxf(mt, tm[c]);
            ms = fmaxf(ms, ss[c] / tau);
            ms1 = fmaxf(ms1, ss[c]);
        }
        float zt = 0.f, zs = 0.f, zs1 = 0.f;
        for (int c = 0; c < C; ++c) {
            zt += __expf(tm[c] - mt);
            zs += __expf(ss[c] / tau - ms);
            zs1 += __expf(ss[c] - ms1);
        }
        float lzs = __logf(zs), lzs1 = __logf(zs1), d = 0.f;
        for (int c = 0; c < C; ++c) {
            float pT = __expf(tm[c] - mt) / zt;
            float ls = ss[c] / tau - ms - lzs;
            d -= pT * ls;
            float pS = __expf(ls);
            float p1 = __expf(ss[c] - ms1 - lzs1);
            float g = (T > 0 ? (pS - pT) / tau : 0.f) + coef * (p1 - (c == y ? 1.f : 0.f));
            dscore[(int64_t)b * C + c] = g / (float)B;
        }
        if (T > 0) l_distill += d;
        l_target += -(ss[y] - ms1 - lzs1);
    }
    red[0][threadIdx.x] = l_distill;
    red[1][threadIdx.x] = l_target;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        losses[0] = red[0][0] / (float)B;
        losses[1] = red[1][0] / (float)B;
    }
}

// embedding-level KD (model_bert.py:277-284, 300-303): one wave per row of the stacked layout
// [B*U history rows | B*C candidate rows | B user rows]; news rows are averaged over U+C, user rows not.
__global__ __launch_bounds__(256) void kd_embed_loss_kernel(const float* __restrict__ S, const float* __restrict__ P,
                                                            const float* __restrict__ tw, float* __restrict__ dS,
                                                            float* __restrict__ dP, float* __restrict__ part, int B,
                                                            int U, int C, int D, int T) {
    const int lane = threadIdx.x & 63;
    const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nh = (int64_t)B * U, nn = (int64_t)B * (U + C), rtot = nn + B;
    if (wv >= rtot) return;
    int b;
    float rowscale;
    if (wv < nh) { b = (int)(wv / U); rowscale = 1.f / (float)(U + C); }
    else if (wv < nn) { b = (int)((wv - nh) / C); rowscale = 1.f / (float)(U + C); }
    else { b = (int)(wv - nn); rowscale = 1.f; }
    float loss = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 s = *(const f32x4*)(S + wv * D + c);
        f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < T; ++i) {
            float wi = tw[(int64_t)b * T + i];
            float ci = wi * rowscale * 2.f / ((float)D * (float)B);
            int64_t off = ((int64_t)i * rtot + wv) * D + c;
            f32x4 df = s - *(const f32x4*)(P + off);
            loss += wi * rowscale * (df[0] * df[0] + df[1] * df[1] + df[2] * df[2] + df[3] * df[3]);
            g += ci * df;
            *(f32x4*)(dP + off) = -ci * df;
        }
        *(f32x4*)(dS + wv * D + c) = g;
    }
    loss = wave_sum(loss);
    if (lane == 0) part[wv] = loss / ((float)D * (float)B);
}

// dvec[row] += dscore[b,c] * user[b]  for candidate rows ; duser[b] += sum_c dscore[b,c] * cand[b,c]
__global__ __launch_bounds__(256) void score_bwd_kernel(const float* __restrict__ vec, const int32_t* __restrict__ cidx,
                                                        const float* __restrict__ user, const float* __restrict__ dscore,
                                                        float* __restrict__ dvec, float* __restrict__ duser, int C, int D) {
    const int b = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) {
        float u = user[(int64_t)b * D + d], acc = 0.f;
        for (int c = 0; c < C; ++c) {
            float g = dscore[(int64_t)b * C + c];
            int64_t row = cidx[(int64_t)b * C + c];
            acc += g * vec[row * D + d];
            dvec[row * D + d] += g * u;
        }
        duser[(int64_t)b * D + d] += acc;
    }
}

#endif   // !TNR_BUILD_F16 (type-independent fp32 kernels are compiled once)

}  // namespace

extern "C" int TNR_NAME(tnr_attpool_fwd)(const void* y, const float* e, int64_t lde, const float* w2, const float* b2, int Q,
                               float* nv, float* alpha, float* den, int64_t n_seq, int L, int H, void* stream) {
    TNR_CHECK_ARG(y && e && w2 && b2 && nv && alpha && den, "tnr_attpool_fwd: null pointer");
    TNR_CHECK_ARG(L >= 1 && L <= 512 && (H % 4) == 0 && Q >= 1 && lde >= Q && n_seq >= 1, "tnr_attpool_fwd: bad shape");
    hipLaunchKernelGGL(attpool_fwd_kernel, dim3((unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, (const bf16*)y, e,
                       lde, w2, b2, Q, nv, alpha, den, L, H);
    TNR_CHECK_LAUNCH("tnr_attpool_fwd");
    return TNR_OK;
}

extern "C" int TNR_NAME(tnr_attpool_bwd)(const void* y, const float* e, int64_t lde, const float* w2, int Q, const float* dnv,
                               const float* alpha, const float* den, void* dy_direct, void* dpre, int64_t lddpre,
                               float* dw2_part, float* db2_part, float* db1_part, int64_t n_seq, int L, int H, void* stream) {
    (void)den;
    TNR_CHECK_ARG(y && e && w2 && dnv && alpha && dy_direct && dpre && dw2_part && db2_part, "tnr_attpool_bwd: null pointer");
    TNR_CHECK_ARG(L >= 1 && L <= 512 && (H % 4) == 0 && Q >= 1 && lde >= Q && lddpre >= Q && n_seq >= 1,
                  "tnr_attpool_bwd: bad shape");
    hipLaunchKernelGGL(attpool_bwd_kernel, dim3((unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, (const bf16*)y, e,
                       lde, w2, Q, dnv, alpha, (bf16*)dy_direct, (bf16*)dpre, lddpre, dw2_part, db2_part, db1_part, L, H);
    TNR_CHECK_LAUNCH("tnr_attpool_bwd");
    return TNR_OK;
}

extern "C" int64_t TNR_NAME(tnr_attpool_long_ws_elems)(int64_t n_seq, int L, int H, int Q, int64_t lddpre) {
    const int64_t nch = (L + AP_CH - 1) / AP_CH;
    const int64_t f = n_seq * nch * H, b = n_seq * L + n_seq * nch * (Q + lddpre + 1);
    return f > b ? f : b;
}

extern "C" int TNR_NAME(tnr_attpool_fwd_long)(const void* y, const float* e, int64_t lde, const float* w2, const float* b2, int Q,
                                    float* nv, float* alpha, float* den, float* ws, int64_t n_seq, int L, int H, void* stream) {
    TNR_CHECK_ARG(y && e && w2 && b2 && nv && alpha && den && ws, "tnr_attpool_fwd_long: null pointer");
    TNR_CHECK_ARG(L >= 1 && L <= 512 && (H % 4) == 0 && Q >= 1 && lde >= Q && n_seq >= 1, "tnr_attpool_fwd_long: bad shape");
    const int64_t n_tok = n_seq * L;
    const unsigned nch = (unsigned)((L + AP_CH - 1) / AP_CH);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attpool_long_score_kernel, dim3((unsigned)((n_tok + 3) / 4)), dim3(256), 0, st, e, lde, w2, b2, Q, alpha, n_tok, L);
    hipLaunchKernelGGL(attpool_long_fwd_part_kernel, dim3((unsigned)n_seq, nch), dim3(256), 0, st, (const bf16*)y, (const float*)alpha, ws, L, H);
    hipLaunchKernelGGL(attpool_long_fwd_fin_kernel, dim3((unsigned)n_seq), dim3(256), 0, st, (const float*)ws, alpha, nv, den, L, H);
    TNR_CHECK_LAUNCH("tnr_attpool_fwd_long");
    return TNR_OK;
}

extern "C" int TNR_NAME(tnr_attpool_bwd_long)(const void* y, const float* e, int64_t lde, const float* w2, int Q, const float* dnv,
                                    const float* alpha, void* dy_direct, void* dpre, int64_t lddpre, float* dw2_part,
                                    float* db2_part, float* db1_part, float* ws, int64_t n_seq, int L, int H, void* stream) {
    TNR_CHECK_ARG(y && e && w2 && dnv && alpha && dy_direct && dpre && dw2_part && db2_part && ws, "tnr_attpool_bwd_long: null pointer");
    TNR_CHECK_ARG(L >= 1 && L <= 512 && (H % 4) == 0 && Q >= 1 && lde >= Q && lddpre >= Q && n_seq >= 1, "tnr_attpool_bwd_long: bad shape");
    const int64_t n_tok = n_seq * L;
    const unsigned nch = (unsigned)((L + AP_CH - 1) / AP_CH);
    hipStream_t st = (hipStream_t)stream;
    float* const part = ws + n_tok;
    hipLaunchKernelGGL(attpool_long_dw_kernel, dim3((unsigned)((n_tok + 3) / 4)), dim3(256), 0, st, (const bf16*)y, dnv, ws, n_tok, L, H);
    hipLaunchKernelGGL(attpool_long_bwd_part_kernel, dim3((unsigned)n_seq, nch), dim3(256), 0, st, e, lde, w2, Q, dnv, alpha, (const float*)ws,
                       (bf16*)dy_direct, (bf16*)dpre, lddpre, part, L, H);
    hipLaunchKernelGGL(attpool_long_bwd_fin_kernel, dim3((unsigned)n_seq), dim3(256), 0, st, (const float*)part, Q, lddpre, (int)nch, dw2_part,
                       db2_part, db1_part);
    TNR_CHECK_LAUNCH("tnr_attpool_bwd_long");
    return TNR_OK;
}

#ifndef TNR_BUILD_F16
extern "C" int tnr_reduce_rows(const float* part, int64_t rows, int64_t stride, int64_t n, float* out, int accumulate,
                               void* stream);

extern "C" int tnr_sgemm(const float* A, int64_t a_rs, int64_t a_cs, int64_t sA, const int32_t* a_idx, const float* B,
                         int64_t b_rs, int64_t b_cs, int64_t sB, float* C, int64_t ldc, int64_t sC, const float* bias,
                         int64_t sBias, int64_t M, int64_t N, int64_t K, int batch, float alpha, float beta,
                         int ksplit, float* part, void* stream) {
    TNR_CHECK_ARG(A && B && C && M >= 1 && N >= 1 && K >= 1 && batch >= 1, "tnr_sgemm: bad argument");
    TNR_CHECK_ARG(a_idx == nullptr, "tnr_sgemm: row gather is done by tnr_gather_rows");
    if (ksplit < 1) ksplit = 1;
    int kchunk = (int)K;
    if (ksplit > 1) {
        TNR_CHECK_ARG(part != nullptr, "tnr_sgemm: split-K needs the partial buffer (ksplit * batch * M * N floats)");
        kchunk = (int)(((K + ksplit - 1) / ksplit + 15) / 16 * 16);      // any multiple of 2 keeps the k pairs of the MFMA aligned
        ksplit = (int)((K + kchunk - 1) / kchunk);
    }
    SgemmArgs g{A, a_rs, a_cs, sA, B, b_rs, b_cs, sB, ksplit > 1 ? part : C, ldc, sC, bias, sBias, (int)M, (int)N, (int)K,
                alpha, ksplit > 1 ? 0.f : beta, ksplit, kchunk, batch, C};
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)(batch * ksplit));
    hipLaunchKernelGGL(sgemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
    TNR_CHECK_LAUNCH("tnr_sgemm");
    if (ksplit > 1) {
        const int64_t total = (int64_t)batch * M * N;
        hipLaunchKernelGGL(sgemm_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                           ksplit, batch, (int)M, (int)N, C, ldc, sC, bias, sBias, alpha, beta);
        TNR_CHECK_LAUNCH("tnr_sgemm/reduce");
    }
    return TNR_OK;
}

extern "C" int tnr_sgemm_group(const tnr_sgemm_problem_t* p, int n, void* stream) {
    TNR_CHECK_ARG(p && n >= 1 && n <= SG_MAXP, "tnr_sgemm_group: 1..%d problems", SG_MAXP);
    SgemmGroup g;
    g.n = n;
    g.start[0] = 0;
    g.rstart[0] = 0;
    for (int i = 0; i < n; ++i) {
        const tnr_sgemm_problem_t& q = p[i];
        TNR_CHECK_ARG(q.A && q.B && q.C && q.M >= 1 && q.N >= 1 && q.K >= 1 && q.batch >= 1, "tnr_sgemm_group: bad problem %d", i);
        int ksplit = q.ksplit < 1 ? 1 : q.ksplit;
        int kchunk = (int)q.K;
        if (ksplit > 1) {
            TNR_CHECK_ARG(q.part != nullptr, "tnr_sgemm_group: split-K needs the partial buffer (problem %d)", i);
            kchunk = (int)(((q.K + ksplit - 1) / ksplit + 15) / 16 * 16);       // the same split rule as tnr_sgemm: identical bits
            ksplit = (int)((q.K + kchunk - 1) / kchunk);
        }
        g.a[i] = SgemmArgs{q.A, q.a_rs, q.a_cs, q.sA, q.B, q.b_rs, q.b_cs, q.sB, ksplit > 1 ? q.part : q.C, q.ldc, q.sC, q.bias, q.sBias,
                           (int)q.M, (int)q.N, (int)q.K, q.alpha, ksplit > 1 ? 0.f : q.beta, ksplit, kchunk, q.batch, q.C};
        g.gx[i] = (int)((q.N + 63) / 64);
        g.gy[i] = (int)((q.M + 63) / 64);
        g.start[i + 1] = g.start[i] + g.gx[i] * g.gy[i] * q.batch * ksplit;
        g.rstart[i + 1] = g.rstart[i] + (ksplit > 1 ? (int64_t)q.batch * q.M * q.N : 0);
        g.alpha[i] = q.alpha;
        g.beta[i] = q.beta;
    }
    for (int i = n; i < SG_MAXP; ++i) { g.start[i + 1] = g.start[n]; g.rstart[i + 1] = g.rstart[n]; g.gx[i] = g.gy[i] = 1; g.a[i] = g.a[0]; g.alpha[i] = g.beta[i] = 0.f; }
    hipLaunchKernelGGL(sgemm_group_kernel, dim3((unsigned)g.start[n]), dim3(256), 0, (hipStream_t)stream, g);
    TNR_CHECK_LAUNCH("tnr_sgemm_group");
    if (g.rstart[n] > 0) {
        hipLaunchKernelGGL(sgemm_group_reduce_kernel, dim3((unsigned)((g.rstart[n] + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g);
        TNR_CHECK_LAUNCH("tnr_sgemm_group/reduce");
    }
    return TNR_OK;
}

// out = [a (na) | b (nb)] int32: the step's news indices, history slots then candidate slots (dataloader.py:129-138 at index level)
__global__ __launch_bounds__(256) void concat_i32_kernel(const int32_t* __restrict__ a, int64_t na, const int32_t* __restrict__ b,
                                                         int64_t nb, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) out[i] = a[i];
    else if (i < na + nb) out[i] = b[i - na];
}

extern "C" int tnr_concat_i32(const int32_t* a, int64_t na, const int32_t* b, int64_t nb, int32_t* out, void* stream) {
    TNR_CHECK_ARG(out && na >= 0 && nb >= 0 && na + nb >= 1 && (a || na == 0) && (b || nb == 0), "tnr_concat_i32: bad argument");
    hipLaunchKernelGGL(concat_i32_kernel, dim3((unsigned)((na + nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, out);
    TNR_CHECK_LAUNCH("tnr_concat_i32");
    return TNR_OK;
}

extern "C" int tnr_gather_rows(const float* tbl, int64_t R, const int32_t* idx, int64_t n_idx, int D, int n_model,
                               float* out, int64_t out_rows, int64_t out_row0, void* stream) {
    TNR_CHECK_ARG(tbl && idx && out && n_idx >= 1 && (D % 4) == 0 && n_model >= 1, "tnr_gather_rows: bad argument");
    int64_t total = n_idx * (D / 4);
    dim3 grid((unsigned)((total + 255) / 256), (unsigned)n_model);
    hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, tbl, R, idx, n_idx, D, out, out_rows,
                       out_row0);
    TNR_CHECK_LAUNCH("tnr_gather_rows");
    return TNR_OK;
}

extern "C" int tnr_segment_sum_rows(const float* src, const int32_t* order, const int32_t* seg, int64_t n_seg, int D,
                                    float* out, void* stream) {
    TNR_CHECK_ARG(src && order && seg && out && n_seg >= 1 && D >= 4 && (D % 4) == 0 && D <= 2048,
                  "tnr_segment_sum_rows: bad argument");
    hipLaunchKernelGGL(segment_sum_rows_kernel, dim3((unsigned)n_seg), dim3(1024), 16 * D * sizeof(float),
                       (hipStream_t)stream, src, order, seg, D, out);
    TNR_CHECK_LAUNCH("tnr_segment_sum_rows");
    return TNR_OK;
}

static int user_shape_ok(int B, int U, int C, int D, int Q) {
    return B >= 1 && U >= 1 && U <= MAXU && C >= 0 && (D % 4) == 0 && D >= 4 && Q >= 4 && (Q % 4) == 0;
}

extern "C" int tnr_user_score_fwd(const float* vec, int64_t R, const int32_t* hidx, const int32_t* cidx,
                                  const float* mask, const float* pad, const float* w1, const float* b1, const float* w2,
                                  const float* b2, int user_log_mask, const float* epre, const float* epad, float* user,
                                  int64_t user_stride,
                                  float* score, float* e, float* alpha, float* den, int n_model, int B, int U, int C, int D,
                                  int Q, void* stream) {
    TNR_CHECK_ARG(vec && hidx && cidx && mask && pad && w1 && b1 && w2 && b2 && user && score && e && alpha && den,
                  "tnr_user_score_fwd: null pointer");
    TNR_CHECK_ARG(user_shape_ok(B, U, C, D, Q) && n_model >= 1 && user_stride >= (int64_t)B * D,
                  "tnr_user_score_fwd: bad shape (U <= %d)", MAXU);
    if (!epre) {                               // fc1 inside the kernel
        size_t lf = sizeof(float) * ((size_t)64 * (D + UF_XTRA) + (size_t)U * Q + MAXU + D);
        TNR_CHECK_ARG((D % 8) == 0 && lf <= 160 * 1024, "tnr_user_score_fwd: fused fc1 needs D %% 8 == 0 and 64 (D + 4) + U Q floats of LDS");
        TNR_ONCE_PER_DEVICE({
            (void)hipFuncSetAttribute((const void*)user_fwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        hipLaunchKernelGGL(user_fwd_fused_kernel, dim3((unsigned)B, (unsigned)n_model), dim3(UF_THREADS), lf, (hipStream_t)stream,
                           vec, R, hidx, cidx, mask, pad, w1, b1, w2, b2, user_log_mask, user, user_stride, score, e, alpha, den,
                           B, U, C, D, Q);
        TNR_CHECK_LAUNCH("tnr_user_score_fwd");
        return TNR_OK;
    }
    size_t lds = sizeof(float) * ((size_t)U * D + (size_t)U * Q + MAXU + D + Q + 4);
    TNR_CHECK_ARG(lds <= 160 * 1024, "tnr_user_score_fwd: U*D + U*Q too large for LDS");
    TNR_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute((const void*)user_score_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(user_score_fwd_kernel, dim3((unsigned)B, (unsigned)n_model), dim3(256), lds, (hipStream_t)stream,
                       vec, R, hidx, cidx, mask, pad, w1, b1, w2, b2, user_log_mask, epre, epad, user, user_stride, score, e, alpha,
                       den, B, U, C, D, Q);
    TNR_CHECK_LAUNCH("tnr_user_score_fwd");
    return TNR_OK;
}

extern "C" int64_t tnr_user_bwd_part_stride(int D, int Q) { return 2 * (int64_t)Q + D + 1; }

extern "C" int tnr_user_bwd_pre(const float* vec, const int32_t* hidx, const float* mask, const float* pad, const float* w2,
                                int user_log_mask, const float* duser, const float* e, const float* alpha, float* hv,
                                float* dpre, float* part, int B, int U, int D, int Q, void* stream) {
    TNR_CHECK_ARG(vec && hidx && mask && pad && w2 && duser && e && alpha && hv && dpre && part, "tnr_user_bwd_pre: null pointer");
    TNR_CHECK_ARG(user_shape_ok(B, U, 0, D, Q), "tnr_user_bwd_pre: bad shape (U <= %d)", MAXU);
    size_t lds = sizeof(float) * ((size_t)U * D + 2 * MAXU);
    TNR_CHECK_ARG(lds <= 160 * 1024, "tnr_user_bwd_pre: too large for LDS");
    TNR_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute((const void*)user_bwd_pre_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(user_bwd_pre_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, vec, hidx, mask, pad, w2,
                       user_log_mask, duser, e, alpha, hv, dpre, part, U, D, Q);
    TNR_CHECK_LAUNCH("tnr_user_bwd_pre");
    return TNR_OK;
}

extern "C" int tnr_user_bwd_post(const float* dhv, const float* alpha, const float* duser, const float* mask,
                                 const int32_t* hidx, int user_log_mask, float* dvec, float* part, int B, int U, int D, int Q,
                                 void* stream) {
    TNR_CHECK_ARG(dhv && alpha && duser && mask && hidx && dvec && part, "tnr_user_bwd_post: null pointer");
    TNR_CHECK_ARG(user_shape_ok(B, U, 0, D, Q), "tnr_user_bwd_post: bad shape (U <= %d)", MAXU);
    hipLaunchKernelGGL(user_bwd_post_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, dhv, alpha, duser, mask,
                       hidx, user_log_mask, dvec, part, U, D, Q);
    TNR_CHECK_LAUNCH("tnr_user_bwd_post");
    return TNR_OK;
}

extern "C" int tnr_user_blend_fwd(const float* vec, int64_t R, const int32_t* hidx, const float* mask, const float* pad,
                                  int user_log_mask, float* hv, int n_model, int B, int U, int D, void* stream) {
    TNR_CHECK_ARG(vec && hidx && mask && pad && hv && n_model >= 1 && B >= 1 && U >= 1 && D >= 4 && (D % 4) == 0,
                  "tnr_user_blend_fwd: bad argument");
    int64_t total = (int64_t)B * U * (D / 4);
    hipLaunchKernelGGL(user_blend_fwd_kernel, dim3((unsigned)((total + 255) / 256), (unsigned)n_model), dim3(256), 0,
                       (hipStream_t)stream, vec, R, hidx, mask, pad, user_log_mask, hv, B, U, D);
    TNR_CHECK_LAUNCH("tnr_user_blend_fwd");
    return TNR_OK;
}

extern "C" int tnr_user_blend_bwd(const float* dhv, const float* mask, const int32_t* hidx, int user_log_mask, float* dvec,
                                  float* pad_part, int64_t part_stride, int B, int U, int D, void* stream) {
    TNR_CHECK_ARG(dhv && mask && hidx && dvec && pad_part && B >= 1 && U >= 1 && D >= 1 && part_stride >= D,
                  "tnr_user_blend_bwd: bad argument");
    hipLaunchKernelGGL(user_blend_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, dhv, mask, hidx,
                       user_log_mask, dvec, pad_part, part_stride, U, D);
    TNR_CHECK_LAUNCH("tnr_user_blend_bwd");
    return TNR_OK;
}

static int nrms_lds_ok(int U, int NH, size_t* lds, bool bwd) {
    *lds = sizeof(float) * ((size_t)2 * U * NH * NRMS_DK + (bwd ? (size_t)2 * U * NH : 0));
    return U >= 1 && NH >= 1 && *lds <= 160 * 1024;
}

extern "C" int tnr_nrms_attn_fwd(const float* qkv, const float* mask, int use_mask, float* ctx, int64_t ctx_rows, int n_model,
                                 int B, int U, int n_heads, void* stream) {
    size_t lds;
    TNR_CHECK_ARG(qkv && ctx && (!use_mask || mask) && n_model >= 1 && B >= 1 && ctx_rows >= (int64_t)B * U,
                  "tnr_nrms_attn_fwd: bad argument");
    TNR_CHECK_ARG(nrms_lds_ok(U, n_heads, &lds, false), "tnr_nrms_attn_fwd: U * heads * 16 too large for LDS");
    TNR_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute((const void*)nrms_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(nrms_attn_fwd_kernel, dim3((unsigned)B, (unsigned)n_model), dim3(256), lds, (hipStream_t)stream, qkv,
                       mask, use_mask, ctx, ctx_rows, B, U, n_heads);
    TNR_CHECK_LAUNCH("tnr_nrms_attn_fwd");
    return TNR_OK;
}

extern "C" int tnr_nrms_attn_bwd(const float* qkv, const float* mask, int use_mask, const float* dctx, float* dqkv, int B,
                                 int U, int n_heads, void* stream) {
    size_t lds;
    TNR_CHECK_ARG(qkv && dctx && dqkv && (!use_mask || mask) && B >= 1, "tnr_nrms_attn_bwd: bad argument");
    TNR_CHECK_ARG(nrms_lds_ok(U, n_heads, &lds, true), "tnr_nrms_attn_bwd: U * heads * 16 too large for LDS");
    TNR_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute((const void*)nrms_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(nrms_attn_bwd_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, qkv, mask, use_mask, dctx,
                       dqkv, U, n_heads);
    TNR_CHECK_LAUNCH("tnr_nrms_attn_bwd");
    return TNR_OK;
}

extern "C" int tnr_kd_score_loss(const float* s_score, const float* t_score, const int64_t* label, float temperature,
                                 float coef, float* tw, float* dscore, float* losses, int B, int C, int T, void* stream) {
    TNR_CHECK_ARG(s_score && label && dscore && losses && (T == 0 || (t_score && tw)), "tnr_kd_score_loss: null pointer");
    TNR_CHECK_ARG(B >= 1 && C >= 1 && C <= 16 && T >= 0 && T <= 16, "tnr_kd_score_loss: need C<=16, T<=16");
    hipLaunchKernelGGL(kd_score_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, s_score, t_score, label,
                       temperature, coef, tw, dscore, losses, B, C, T);
    TNR_CHECK_LAUNCH("tnr_kd_score_loss");
    return TNR_OK;
}



extern "C" int tnr_kd_embed_loss(const float* S, const float* P, const float* tw, float* loss, float* dS, float* dP,
                                 float* part, int B, int U, int C, int D, int T, void* stream) {
    TNR_CHECK_ARG(S && P && tw && loss && dS && dP && part, "tnr_kd_embed_loss: null pointer");
    TNR_CHECK_ARG(B >= 1 && U >= 0 && C >= 1 && (D % 4) == 0 && T >= 1, "tnr_kd_embed_loss: bad shape");
    int64_t waves = (int64_t)B * (U + C + 1);
    hipLaunchKernelGGL(kd_embed_loss_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, P, tw,
                       dS, dP, part, B, U, C, D, T);
    TNR_CHECK_LAUNCH("tnr_kd_embed_loss");
    return tnr_reduce_rows(part, waves, 1, 1, loss, 0, stream);
}

extern "C" int tnr_score_bwd(const float* vec, const int32_t* cidx, const float* user, const float* dscore, float* dvec,
                             float* duser, int B, int C, int D, void* stream) {
    TNR_CHECK_ARG(vec && cidx && user && dscore && dvec && duser && B >= 1 && C >= 1 && D >= 1, "tnr_score_bwd: bad argument");
    hipLaunchKernelGGL(score_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, vec, cidx, user, dscore, dvec,
                       duser, C, D);
    TNR_CHECK_LAUNCH("tnr_score_bwd");
    return TNR_OK;
}

#endif   // !TNR_BUILD_F16
