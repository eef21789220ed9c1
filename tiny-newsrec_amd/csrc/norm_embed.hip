// HBM-bound row kernels: embedding gather + LayerNorm, LayerNorm fwd/bwd, rel-pos table, column sums.
// LayerNorm forward / backward: HALF a wave per row of H = 256*V elements, each lane owning V runs of 8 consecutive elements, so that
// every access of the 16-bit rows is 16 bytes wide (512 contiguous bytes per half wave and instruction; the 8-byte form of
// rounds 1-2 ran at 0.50-0.60 of the HBM rate).  The embedding kernel keeps one wave per row: its rows are fp32 (16-byte accesses
// already).
#include "common.h"
#include "dropout.h"

namespace {

template <int V>
__device__ __forceinline__ void row_stats(const float (&x)[V][4], int H, float& mean, float& rstd, float eps) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += x[v][r];
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float d = x[v][r] - mean;
            q += d * d;
        }
    rstd = rsqrtf(wave_sum(q) / (float)H + eps);
}

// tnlrv3/modeling.py:153-178 (word + pos + type0 -> LN) fused with the mask of :446-454
template <int V, typename TokT>
__global__ __launch_bounds__(256) void embed_ln_kernel(const TokT* __restrict__ tok, const int32_t* __restrict__ nidx,
                                                       int64_t n_tok, int L,
                                                       const float* __restrict__ word, const float* __restrict__ pos,
                                                       const float* __restrict__ type0, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, bf16* __restrict__ out,
                                                       float* __restrict__ mask_add, TnrDrop drop,
                                                       const int32_t* __restrict__ pos_ids) {
    const int H = 256 * V;
    const int lane = threadIdx.x & 63;
    int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tok) return;
    int64_t n = t / L;
    int i = (int)(t - n * L);
    const int64_t trow = nidx ? (int64_t)nidx[n] : n;       // news index -> row of the resident token table
    int64_t id = (int64_t)tok[trow * 2 * L + i];
    if (lane == 0) {
        const int Lr = (L + 31) & ~31;                       // mask rows are padded to a multiple of 32 keys
        int64_t mk = (int64_t)tok[trow * 2 * L + L + i];
        mask_add[n * Lr + i] = (1.0f - (float)mk) * -10000.0f;
        if (i == 0)
            for (int j = L; j < Lr; ++j) mask_add[n * Lr + j] = -1e30f;
    }
    // position row: the token's index (BERT / UniLM, tnlrv3/modeling.py:164-167), or a per-token id table laid out like the
    // token table (RoBERTa: cumulative count of non-pad tokens + padding_idx, PLM-NR's --model_type roberta)
    const int pi = pos_ids ? pos_ids[trow * L + i] : i;
    float x[V][4];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        int c = v * 256 + lane * 4;
        f32x4 a = *(const f32x4*)(word + id * H + c);
        f32x4 b = *(const f32x4*)(pos + (int64_t)pi * H + c);
        f32x4 d = *(const f32x4*)(type0 + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[v][r] = a[r] + b[r] + d[r];
    }
    float mean, rstd;
    row_stats<V>(x, H, mean, rstd, eps);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        int c = v * 256 + lane * 4;
        f32x4 gm = *(const f32x4*)(gamma + c);
        f32x4 bt = *(const f32x4*)(beta + c);
        float dm[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop.thresh) tnr_drop4(drop, (uint64_t)t * H + c, dm);          // tnlrv3/modeling.py:177
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(((x[v][r] - mean) * rstd * gm[r] + bt[r]) * dm[r]);
        *(bf16x4*)(out + t * H + c) = o;
    }
}

__device__ __forceinline__ float half_sum(float v) {      // over the 32 lanes of this lane's half wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int V>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ xin, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, bf16* __restrict__ y,
                                                     float* __restrict__ stats, int64_t M) {
    const int H = 256 * V;
    const int hl = threadIdx.x & 31;
    int64_t m = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (m >= M) return;
    float x[V][8];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const bf16x8 a = *(const bf16x8*)(xin + m * H + v * 256 + hl * 8);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            x[v][r] = (float)a[r];
            s += x[v][r];
        }
    }
    const float mean = half_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float d = x[v][r] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(half_sum(q) / (float)H + eps);
    if (stats && hl == 0) {
        stats[m * 2] = mean;
        stats[m * 2 + 1] = rstd;
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int c = v * 256 + hl * 8;
        const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
        const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = (bf16)((x[v][r] - mean) * rstd * g0[r] + b0[r]);
            o[4 + r] = (bf16)((x[v][4 + r] - mean) * rstd * g1[r] + b1[r]);
        }
        *(bf16x8*)(y + m * H + c) = o;
    }
}

// dx = rstd * (dxh - mean(dxh) - xh * mean(dxh*xh)), dxh = dy*gamma ; per-block partial dgamma/dbeta
constexpr int LNB_ROWS = 128;  // rows per block (8 half waves x 16 rows); short inputs use 32 so that every CU gets work
static inline int lnb_rows(int64_t M) { return M >= 32768 ? LNB_ROWS : 32; }
static inline int64_t lnb_blocks(int64_t M) { return (M + lnb_rows(M) - 1) / lnb_rows(M); }
template <int V>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ xin,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     bf16* __restrict__ dx, float* __restrict__ part, int64_t M,
                                                     int rows_per_block, bf16* __restrict__ dxm, TnrDrop drop) {
    const int H = 256 * V;
    __shared__ float red[4][3][256 * V];
    const int lane = threadIdx.x & 63, hl = lane & 31, w = threadIdx.x >> 6, hw = threadIdx.x >> 5;
    float gm[V][8], dg[V][8], db[V][8], dxs[V][8];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const f32x4 t0 = *(const f32x4*)(gamma + v * 256 + hl * 8), t1 = *(const f32x4*)(gamma + v * 256 + hl * 8 + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { gm[v][r] = t0[r]; gm[v][4 + r] = t1[r]; }
#pragma unroll
        for (int r = 0; r < 8; ++r) { dg[v][r] = 0.f; db[v][r] = 0.f; dxs[v][r] = 0.f; }
    }
    for (int it = 0; it < rows_per_block / 8; ++it) {
        const int64_t m = (int64_t)blockIdx.x * rows_per_block + it * 8 + hw;
        if (m >= M) break;
        const float mean = stats[m * 2], rstd = stats[m * 2 + 1];
        float xh[V][8], g[V][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const bf16x8 a = *(const bf16x8*)(xin + m * H + v * 256 + hl * 8);
            const bf16x8 d = *(const bf16x8*)(dy + m * H + v * 256 + hl * 8);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float dyv = (float)d[r];
                xh[v][r] = ((float)a[r] - mean) * rstd;
                g[v][r] = dyv * gm[v][r];
                s1 += g[v][r];
                s2 += g[v][r] * xh[v][r];
                dg[v][r] += dyv * xh[v][r];
                db[v][r] += dyv;
            }
        }
        s1 = half_sum(s1) / (float)H;
        s2 = half_sum(s2) / (float)H;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            bf16x8 o;
            if (dxm) {
                // the Linear in front of this LayerNorm was followed by dropout (BertSelfOutput / BertOutput): its output
                // gradient is dx * mask / (1 - p) (second output, what its wgrad / dgrad / bias gradient consume), the
                // residual branch takes dx itself
                float dm[8];
                tnr_drop8(drop, ((uint64_t)m * H + v * 256 + hl * 8) >> 3, dm);
                bf16x8 om;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float t = rstd * (g[v][r] - s1 - xh[v][r] * s2);
                    o[r] = (bf16)t;
                    om[r] = (bf16)(t * dm[r]);
                    dxs[v][r] += (float)om[r];
                }
                *(bf16x8*)(dxm + m * H + v * 256 + hl * 8) = om;
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    o[r] = (bf16)(rstd * (g[v][r] - s1 - xh[v][r] * s2));
                    dxs[v][r] += (float)o[r];          // the rounded value the wgrad kernels will see
                }
            }
            *(bf16x8*)(dx + m * H + v * 256 + hl * 8) = o;
        }
    }
    if (part == nullptr) return;
    // the two half waves of a wave own the same columns: combine them, then the four waves through LDS (fixed order)
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float a = dg[v][r] + __shfl_xor(dg[v][r], 32, 64);
            const float b = db[v][r] + __shfl_xor(db[v][r], 32, 64);
            const float c = dxs[v][r] + __shfl_xor(dxs[v][r], 32, 64);
            if (lane < 32) {
                red[w][0][v * 256 + hl * 8 + r] = a;
                red[w][1][v * 256 + hl * 8 + r] = b;
                red[w][2][v * 256 + hl * 8 + r] = c;
            }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * H; c += 256) {
        int k = c / H, h = c - k * H;
        part[(int64_t)blockIdx.x * 3 * H + c] = red[0][k][h] + red[1][k][h] + red[2][k][h] + red[3][k][h];
    }
}

// fixed-order sum over rows of a (rows, stride) fp32 matrix.  Wide outputs: block = 64 columns x 4 row lanes
// (each lane a strided subsequence, combined in a fixed order).  Narrow outputs (n < 64): one block per column.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int64_t rows, int64_t stride,
                                                          int64_t n, float* __restrict__ out, int accumulate) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + c;
    float s = 0.f;
    if (i < n)
        for (int64_t r = rl; r < rows; r += 4) s += part[r * stride + i];
    red[rl][c] = s;
    __syncthreads();
    if (rl == 0 && i < n) {
        float t = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        out[i] = accumulate ? out[i] + t : t;
    }
}
// many reductions in one launch: one descriptor per BLOCK on the device, 6 int64
// {src ptr, rows, row stride (floats), ncols <= 64, dst ptr, accumulate | scale bits << 32}: dst[c] (+)= scale * sum_r src[r*stride + c]
// (scale = the fp32 whose bit pattern sits in the upper half of the last word; upper half 0 means 1.0).
// Fixed order (4 row lanes, each a strided subsequence, combined as ((0+1)+(2+3))).  The host builds two tables
// per batch: level 1 sums row chunks IN PLACE (dst = first row of the chunk), level 2 sums the chunk rows.
__global__ __launch_bounds__(256) void reduce_multi_kernel(const int64_t* __restrict__ desc) {
    __shared__ float red[4][64];
    const int64_t* d = desc + (int64_t)blockIdx.x * 6;
    const float* src = (const float*)d[0];
    const int64_t rows = d[1], stride = d[2];
    const int ncols = (int)d[3];
    float* dst = (float*)d[4];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f;
    if (c < ncols) {
        int64_t r = rl;
        for (; r + 4 < rows; r += 8) {               // two independent chains: more loads in flight
            s0 += src[r * stride + c];
            s1 += src[(r + 4) * stride + c];
        }
        if (r < rows) s0 += src[r * stride + c];
    }
    red[rl][c] = s0 + s1;
    __syncthreads();
    if (rl == 0 && c < ncols) {
        float t = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        const unsigned sb = (unsigned)((uint64_t)d[5] >> 32);
        if (sb) t *= __uint_as_float(sb);
        dst[c] = (d[5] & 1) ? dst[c] + t : t;
    }
}

// first level for tall inputs: chunk y sums its rows IN PLACE into its first row (each block only touches its own
// 64 columns of its own chunk), so that the second level reads `chunks` rows instead of `rows`
__global__ __launch_bounds__(256) void reduce_rows_chunk_kernel(float* __restrict__ part, int64_t rows, int64_t stride,
                                                                int64_t n, int64_t chunk) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + c;
    const int64_t r0 = (int64_t)blockIdx.y * chunk;
    const int64_t r1 = r0 + chunk < rows ? r0 + chunk : rows;
    float s = 0.f;
    if (i < n)
        for (int64_t r = r0 + rl; r < r1; r += 4) s += part[r * stride + i];
    red[rl][c] = s;
    __syncthreads();
    if (rl == 0 && i < n) part[r0 * stride + i] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ __launch_bounds__(256) void reduce_rows_narrow_kernel(const float* __restrict__ part, int64_t rows,
                                                                 int64_t stride, float* __restrict__ out, int accumulate) {
    __shared__ float red[256];
    const int64_t i = blockIdx.x;
    float s = 0.f;
    for (int64_t r = threadIdx.x; r < rows; r += 256) s += part[r * stride + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = accumulate ? out[i] + red[0] : red[0];
}

// column sums: block = 256 columns (64 threads x 4) x 4 row lanes over `rows_per_block` rows: 512 for tall inputs, 64 for short
// ones (a 1792-row input on 512-row blocks was 4 workgroups walking 128 dependent loads each: 33 us for 1.8 MB)
constexpr int CS_ROWS = 512;
static inline int cs_rows(int64_t M) { return M >= 32768 ? CS_ROWS : 64; }
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ Xb, int64_t ldx, int64_t M, int64_t N,
                                                     float* __restrict__ partb, int64_t sX, int64_t sPart, int rows_per_block) {
    const T* X = Xb + (int64_t)blockIdx.z * sX;
    float* part = partb + (int64_t)blockIdx.z * sPart;
    // thread handles 4 consecutive columns; 64 threads across 256 columns, 4 row groups
    __shared__ float red[4][256];
    const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 256 + cg * 4;
    const int64_t m0 = (int64_t)blockIdx.y * rows_per_block;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        int64_t mend = m0 + rows_per_block < M ? m0 + rows_per_block : M;
        for (int64_t m = m0 + rg; m < mend; m += 4) {
            if constexpr (sizeof(T) == 2) {
                bf16x4 a = *(const bf16x4*)((const bf16*)X + m * ldx + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] += (float)a[r];
            } else {
                f32x4 a = *(const f32x4*)((const float*)X + m * ldx + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] += a[r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[rg][cg * 4 + r] = s[r];
    __syncthreads();
    int t = threadIdx.x;
    int64_t col = (int64_t)blockIdx.x * 256 + t;
    if (col < N) part[(int64_t)blockIdx.y * gridDim.z * N + col] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// tnlrv3/modeling.py:345-373 bucket (integer edges, see oracle) + the Linear(32->A) lookup of :462-463
__device__ __forceinline__ int relpos_bucket(int rel) {
    int n = rel < 0 ? -rel : rel;
    int b;
    if (n < 8) b = n;
    else if (n < 12) b = 8;
    else if (n < 16) b = 9;
    else if (n < 23) b = 10;
    else if (n < 32) b = 11;
    else if (n < 46) b = 12;
    else if (n < 64) b = 13;
    else if (n < 91) b = 14;
    else b = 15;
    return (rel > 0 ? 16 : 0) + b;
}
__global__ void relpos_kernel(const float* __restrict__ weight, int A, int L, int Lr, float* __restrict__ table) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)A * Lr * Lr) return;
    int a = (int)(idx / ((int64_t)Lr * Lr));
    int rem = (int)(idx - (int64_t)a * Lr * Lr);
    int i = rem / Lr, j = rem - i * Lr;
    table[idx] = (i < L && j < L) ? weight[a * 32 + relpos_bucket(j - i)] : 0.f;
}

__global__ void cast_f2b_kernel(const float* __restrict__ s, bf16* __restrict__ d, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (bf16)s[i];
}
__global__ void cast_b2f_kernel(const bf16* __restrict__ s, float* __restrict__ d, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (float)s[i];
}


// NewsEncoder pooling other than 'att' (model_bert.py:130-135): 'cls' = hidden state of token 0, otherwise the mean
// over ALL L positions (padding included, as torch.mean(word_vecs, dim=1)).  One workgroup per sequence.
__global__ __launch_bounds__(256) void pool_fwd_kernel(const bf16* __restrict__ y, float* __restrict__ nv, int L, int H,
                                                       int mean) {
    const int64_t n = blockIdx.x;
    for (int c = threadIdx.x * 4; c < H; c += 1024) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int rows = mean ? L : 1;
        for (int i = 0; i < rows; ++i) {
            bf16x4 v = *(const bf16x4*)(y + (n * L + i) * H + c);
            a += (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        }
        if (mean) a *= 1.0f / (float)L;
        *(f32x4*)(nv + n * H + c) = a;
    }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dnv, bf16* __restrict__ dy, int L, int H,
                                                       int mean) {
    const int64_t n = blockIdx.x;
    for (int c = threadIdx.x * 4; c < H; c += 1024) {
        f32x4 g = *(const f32x4*)(dnv + n * H + c);
        if (mean) g *= 1.0f / (float)L;
        bf16x4 o = (bf16x4){(bf16)g[0], (bf16)g[1], (bf16)g[2], (bf16)g[3]};
        bf16x4 zero = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        for (int i = 0; i < L; ++i) *(bf16x4*)(dy + (n * L + i) * H + c) = (mean || i == 0) ? o : zero;
    }
}

}  // namespace

#ifndef TNR_BUILD_F16
extern "C" int tnr_relpos_table(const float* weight, int A, int L, float* table, void* stream) {
    TNR_CHECK_ARG(weight && table && A >= 1 && L >= 1 && L <= 512, "tnr_relpos_table: need 1<=L<=512");
    const int Lr = (L + 31) / 32 * 32;        // table is (A, Lr, Lr): (A,32,32) for the fused short kernel
    hipLaunchKernelGGL(relpos_kernel, dim3((unsigned)(((int64_t)A * Lr * Lr + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       weight, A, L, Lr, table);
    TNR_CHECK_LAUNCH("tnr_relpos_table");
    return TNR_OK;
}

#endif

extern "C" int TNR_NAME(tnr_embed_ln_fwd_do)(const int64_t* tok, int64_t n_seq, int L, int H, const float* word, const float* pos,
                                   const float* type0, const float* gamma, const float* beta, float eps, void* out,
                                   float* mask_add, const tnr_dropout_t* drop, const int32_t* pos_ids, void* stream) {
    TnrDrop dd;
    if (int rc = tnr_make_drop(drop, &dd, "tnr_embed_ln_fwd")) return rc;
    TNR_CHECK_ARG(tok && word && pos && type0 && gamma && beta && out && mask_add, "tnr_embed_ln_fwd: null pointer");
    TNR_CHECK_ARG(L >= 1 && L <= 512 && n_seq >= 1, "tnr_embed_ln_fwd: need 1<=L<=512");
    TNR_CHECK_ARG(H == 768 || H == 256 || H == 512 || H == 1024, "tnr_embed_ln_fwd: H must be 256/512/768/1024");
    int64_t n_tok = n_seq * L;
    dim3 grid((unsigned)((n_tok + 3) / 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(V) hipLaunchKernelGGL((embed_ln_kernel<V, int64_t>), grid, blk, 0, st, tok, (const int32_t*)nullptr, n_tok, L, word, pos, type0, gamma, beta, eps, (bf16*)out, mask_add, dd, pos_ids)
    switch (H / 256) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
    TNR_CHECK_LAUNCH("tnr_embed_ln_fwd");
    return TNR_OK;
}
extern "C" int TNR_NAME(tnr_embed_ln_fwd)(const int64_t* tok, int64_t n_seq, int L, int H, const float* word, const float* pos,
                                const float* type0, const float* gamma, const float* beta, float eps, void* out,
                                float* mask_add, void* stream) {
    return TNR_NAME(tnr_embed_ln_fwd_do)(tok, n_seq, L, H, word, pos, type0, gamma, beta, eps, out, mask_add, nullptr, nullptr, stream);
}

extern "C" int TNR_NAME(tnr_embed_ln_fwd_indexed_do)(const int32_t* news_combined, const int32_t* nidx, int64_t n_seq, int L, int H,
                                           const float* word, const float* pos, const float* type0, const float* gamma,
                                           const float* beta, float eps, void* out, float* mask_add,
                                           const tnr_dropout_t* drop, const int32_t* pos_ids, void* stream) {
    TnrDrop dd;
    if (int rc = tnr_make_drop(drop, &dd, "tnr_embed_ln_fwd_indexed")) return rc;
    TNR_CHECK_ARG(news_combined && nidx && word && pos && type0 && gamma && beta && out && mask_add,
                  "tnr_embed_ln_fwd_indexed: null pointer");
    TNR_CHECK_ARG(L >= 1 && L <= 512 && n_seq >= 1, "tnr_embed_ln_fwd_indexed: need 1<=L<=512");
    TNR_CHECK_ARG(H == 768 || H == 256 || H == 512 || H == 1024, "tnr_embed_ln_fwd_indexed: H must be 256/512/768/1024");
    int64_t n_tok = n_seq * L;
    dim3 grid((unsigned)((n_tok + 3) / 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(V) hipLaunchKernelGGL((embed_ln_kernel<V, int32_t>), grid, blk, 0, st, news_combined, nidx, n_tok, L, word, pos, type0, gamma, beta, eps, (bf16*)out, mask_add, dd, pos_ids)
    switch (H / 256) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
    TNR_CHECK_LAUNCH("tnr_embed_ln_fwd_indexed");
    return TNR_OK;
}
extern "C" int TNR_NAME(tnr_embed_ln_fwd_indexed)(const int32_t* news_combined, const int32_t* nidx, int64_t n_seq, int L, int H,
                                        const float* word, const float* pos, const float* type0, const float* gamma,
                                        const float* beta, float eps, void* out, float* mask_add, void* stream) {
    return TNR_NAME(tnr_embed_ln_fwd_indexed_do)(news_combined, nidx, n_seq, L, H, word, pos, type0, gamma, beta, eps, out, mask_add,
                                                 nullptr, nullptr, stream);
}

extern "C" int TNR_NAME(tnr_pool_fwd)(const void* y, float* nv, int64_t n_seq, int L, int H, int mean, void* stream) {
    TNR_CHECK_ARG(y && nv && n_seq >= 1 && L >= 1 && H >= 4 && (H % 4) == 0, "tnr_pool_fwd: bad argument");
    hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, (const bf16*)y, nv, L, H,
                       mean);
    TNR_CHECK_LAUNCH("tnr_pool_fwd");
    return TNR_OK;
}

extern "C" int TNR_NAME(tnr_pool_bwd)(const float* dnv, void* dy, int64_t n_seq, int L, int H, int mean, void* stream) {
    TNR_CHECK_ARG(dnv && dy && n_seq >= 1 && L >= 1 && H >= 4 && (H % 4) == 0, "tnr_pool_bwd: bad argument");
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, dnv, (bf16*)dy, L, H, mean);
    TNR_CHECK_LAUNCH("tnr_pool_bwd");
    return TNR_OK;
}

extern "C" int TNR_NAME(tnr_ln_fwd)(const void* x, const float* gamma, const float* beta, float eps, void* y, float* stats,
                          int64_t M, int H, void* stream) {
    TNR_CHECK_ARG(x && gamma && beta && y && M >= 1, "tnr_ln_fwd: null pointer");
    TNR_CHECK_ARG(H == 768 || H == 256 || H == 512 || H == 1024, "tnr_ln_fwd: H must be 256/512/768/1024");
    dim3 grid((unsigned)((M + 7) / 8)), blk(256);        // half a wave per row
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(V) hipLaunchKernelGGL(ln_fwd_kernel<V>, grid, blk, 0, st, (const bf16*)x, gamma, beta, eps, (bf16*)y, stats, M)
    switch (H / 256) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
    TNR_CHECK_LAUNCH("tnr_ln_fwd");
    return TNR_OK;
}

#ifndef TNR_BUILD_F16
// workspace for any row count up to M (short inputs use smaller blocks, i.e. more partial rows)
extern "C" int64_t tnr_ln_bwd_part_elems(int64_t M, int H) {
    int64_t nb = lnb_blocks(M), nb_short = lnb_blocks(M < 32767 ? M : 32767);
    return (nb > nb_short ? nb : nb_short) * 3 * H;
}
extern "C" int64_t tnr_ln_bwd_blocks(int64_t M) { return lnb_blocks(M); }

extern "C" int tnr_reduce_rows(const float* part, int64_t rows, int64_t stride, int64_t n, float* out, int accumulate,
                               void* stream) {
    TNR_CHECK_ARG(part && out && rows >= 1 && n >= 1, "tnr_reduce_rows: bad argument");
    const int64_t colblk = (n + 63) / 64;
    if (n < 64 && rows >= 256) {
        hipLaunchKernelGGL(reduce_rows_narrow_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, part, rows,
                           stride, out, accumulate);
    } else if (rows >= 128 && colblk < 512) {
        // two levels: enough workgroups to stream the partials at HBM rate; `part` is clobbered (it is scratch)
        int64_t chunks = 1024 / colblk;
        if (chunks > rows / 16) chunks = rows / 16;
        if (chunks < 2) chunks = 2;
        int64_t chunk = (rows + chunks - 1) / chunks;
        chunks = (rows + chunk - 1) / chunk;
        hipLaunchKernelGGL(reduce_rows_chunk_kernel, dim3((unsigned)colblk, (unsigned)chunks), dim3(256), 0,
                           (hipStream_t)stream, (float*)part, rows, stride, n, chunk);
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)colblk), dim3(256), 0, (hipStream_t)stream, part, chunks,
                           chunk * stride, n, out, accumulate);
    } else {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)colblk), dim3(256), 0, (hipStream_t)stream, part, rows,
                           stride, n, out, accumulate);
    }
    TNR_CHECK_LAUNCH("tnr_reduce_rows");
    return TNR_OK;
}

#else
extern "C" int tnr_reduce_rows(const float* part, int64_t rows, int64_t stride, int64_t n, float* out, int accumulate,
                               void* stream);
#endif

#ifndef TNR_BUILD_F16
// the multipliers of a dropout site as fp32 (tests: statistics, bit equality with oracle/dropout_oracle.py)
__global__ void dropout_mask_rows_kernel(float* __restrict__ out, int64_t n4, TnrDrop d) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float m[4];
    tnr_drop4(d, (uint64_t)i * 4, m);
    *(f32x4*)(out + i * 4) = (f32x4){m[0], m[1], m[2], m[3]};
}
// attention probabilities (pairs, L, L): thread = (pair, query, key group of 4) through the row accessor, or (cols != 0) the
// column accessor (four queries of one key), so both device paths are pinned
__global__ void dropout_mask_probs_kernel(float* __restrict__ out, int64_t pairs, int L, int Lr, int cols, TnrDrop d) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int g = Lr / 4;
    if (i >= pairs * Lr * g) return;
    const int64_t pair = i / ((int64_t)Lr * g);
    const int rem = (int)(i - pair * Lr * g);
    float m[4];
    if (!cols) {
        const int q = rem / g, k0 = (rem - q * g) * 4;
        tnr_drop_prob_row(d, (uint64_t)pair, g, q, k0, m);
        for (int e = 0; e < 4; ++e)
            if (q < L && k0 + e < L) out[(pair * L + q) * L + k0 + e] = m[e];
    } else {
        const int k = rem / g, q0 = (rem - k * g) * 4;
        tnr_drop_prob_col(d, (uint64_t)pair, g, q0, k, m);
        for (int e = 0; e < 4; ++e)
            if (k < L && q0 + e < L) out[(pair * L + q0 + e) * L + k] = m[e];
    }
}
extern "C" int tnr_dropout_mask(const tnr_dropout_t* drop, int64_t rows, int64_t cols, float* out, void* stream) {
    TnrDrop dd;
    if (int rc = tnr_make_drop(drop, &dd, "tnr_dropout_mask")) return rc;
    TNR_CHECK_ARG(out && rows >= 1 && cols >= 4 && (cols % 4) == 0, "tnr_dropout_mask: need cols %% 4 == 0");
    int64_t n4 = rows * cols / 4;
    hipLaunchKernelGGL(dropout_mask_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n4, dd);
    TNR_CHECK_LAUNCH("tnr_dropout_mask");
    return TNR_OK;
}
extern "C" int tnr_dropout_mask_probs(const tnr_dropout_t* drop, int64_t pairs, int L, int by_columns, float* out, void* stream) {
    TnrDrop dd;
    if (int rc = tnr_make_drop(drop, &dd, "tnr_dropout_mask_probs")) return rc;
    TNR_CHECK_ARG(out && pairs >= 1 && L >= 1 && L <= 512, "tnr_dropout_mask_probs: bad argument");
    const int Lr = (L + 31) / 32 * 32;
    int64_t n = pairs * Lr * (Lr / 4);
    hipLaunchKernelGGL(dropout_mask_probs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, pairs, L,
                       Lr, by_columns, dd);
    TNR_CHECK_LAUNCH("tnr_dropout_mask_probs");
    return TNR_OK;
}
__global__ void scale_inplace_kernel(float* __restrict__ x, int64_t n, float s) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= s;
}
extern "C" int tnr_scale_inplace(float* x, int64_t n, float s, void* stream) {
    TNR_CHECK_ARG(x && n >= 1, "tnr_scale_inplace: bad argument");
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, s);
    TNR_CHECK_LAUNCH("tnr_scale_inplace");
    return TNR_OK;
}
extern "C" int tnr_reduce_multi(const int64_t* desc, int n_blocks, void* stream) {
    TNR_CHECK_ARG(desc && n_blocks >= 1, "tnr_reduce_multi: bad argument");
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, desc);
    TNR_CHECK_LAUNCH("tnr_reduce_multi");
    return TNR_OK;
}
#endif

extern "C" int TNR_NAME(tnr_ln_bwd_do)(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                             float* dgamma, float* dbeta, float* dxsum, float* part, int64_t M, int H, void* dxm,
                             const tnr_dropout_t* drop, void* stream) {
    // dgamma == dbeta == dxsum == NULL with part != NULL: partials only, the caller reduces them (tnr_reduce_multi)
    TnrDrop dd;
    if (int rc = tnr_make_drop(drop, &dd, "tnr_ln_bwd")) return rc;
    TNR_CHECK_ARG(dxm || !dd.thresh, "tnr_ln_bwd: an active dropout site needs the masked second output");
    TNR_CHECK_ARG(dy && x && stats && gamma && dx && M >= 1, "tnr_ln_bwd: null pointer");
    TNR_CHECK_ARG(H == 768 || H == 256 || H == 512 || H == 1024, "tnr_ln_bwd: H must be 256/512/768/1024");
    TNR_CHECK_ARG(!(dgamma || dbeta || dxsum) || part, "tnr_ln_bwd: part workspace required for dgamma/dbeta/dxsum");
    int64_t nblk = lnb_blocks(M);
    const int rows = lnb_rows(M);
    dim3 grid((unsigned)nblk), blk(256);
    hipStream_t st = (hipStream_t)stream;
    float* p = part;
#define LAUNCH(V) hipLaunchKernelGGL(ln_bwd_kernel<V>, grid, blk, 0, st, (const bf16*)dy, (const bf16*)x, stats, gamma, (bf16*)dx, p, M, rows, (bf16*)dxm, dd)
    switch (H / 256) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
    TNR_CHECK_LAUNCH("tnr_ln_bwd");
    if (dgamma && dbeta == dgamma + H) {       // adjacent in the flat gradient buffer: one reduction
        int rc = tnr_reduce_rows(part, nblk, 3 * H, 2 * H, dgamma, 0, stream); if (rc) return rc;
    } else {
        if (dgamma) { int rc = tnr_reduce_rows(part, nblk, 3 * H, H, dgamma, 0, stream); if (rc) return rc; }
        if (dbeta) { int rc = tnr_reduce_rows(part + H, nblk, 3 * H, H, dbeta, 0, stream); if (rc) return rc; }
    }
    if (dxsum) { int rc = tnr_reduce_rows(part + 2 * H, nblk, 3 * H, H, dxsum, 0, stream); if (rc) return rc; }
    return TNR_OK;
}
extern "C" int TNR_NAME(tnr_ln_bwd)(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                          float* dgamma, float* dbeta, float* dxsum, float* part, int64_t M, int H, void* stream) {
    return TNR_NAME(tnr_ln_bwd_do)(dy, x, stats, gamma, dx, dgamma, dbeta, dxsum, part, M, H, nullptr, nullptr, stream);
}

#ifndef TNR_BUILD_F16
// monotone in M: a workspace sized for M rows serves every call with fewer rows
extern "C" int64_t tnr_colsum_part_elems(int64_t M, int64_t N) {
    int64_t tall = (M + CS_ROWS - 1) / CS_ROWS, small = (M + 63) / 64;
    if (small > 32768 / 64) small = 32768 / 64;
    return (tall > small ? tall : small) * N;
}
#endif

extern "C" int TNR_NAME(tnr_colsum_batched)(const void* X, int64_t ldx, int64_t sX, int dtype, int64_t M, int64_t N, int batch,
                                  float* out, float* part, int accumulate, void* stream) {
    TNR_CHECK_ARG(X && out && part && M >= 1 && N >= 4 && (N % 4) == 0 && (ldx % 4) == 0 && batch >= 1, "tnr_colsum: bad argument");
    TNR_CHECK_ARG(dtype == TNR_BF16 || dtype == TNR_F16 || dtype == TNR_F32, "tnr_colsum: dtype");
    const int rpb = cs_rows(M);
    int64_t nby = (M + rpb - 1) / rpb;
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)nby, (unsigned)batch), blk(256);
    // partials laid out (nby, batch, N) so that ONE row reduction yields out (batch, N)
    if (dtype != TNR_F32)      // the 16-bit type of this build
        hipLaunchKernelGGL(colsum_kernel<bf16>, grid, blk, 0, (hipStream_t)stream, (const bf16*)X, ldx, M, N, part, sX, N, rpb);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, blk, 0, (hipStream_t)stream, (const float*)X, ldx, M, N, part, sX, N, rpb);
    TNR_CHECK_LAUNCH("tnr_colsum");
    return tnr_reduce_rows(part, nby, (int64_t)batch * N, (int64_t)batch * N, out, accumulate, stream);
}

extern "C" int TNR_NAME(tnr_colsum)(const void* X, int64_t ldx, int dtype, int64_t M, int64_t N, float* out, float* part,
                          int accumulate, void* stream) {
    return TNR_NAME(tnr_colsum_batched)(X, ldx, 0, dtype, M, N, 1, out, part, accumulate, stream);
}

extern "C" int TNR_NAME(tnr_cast_f32_to_bf16)(const float* src, void* dst, int64_t n, void* stream) {
    TNR_CHECK_ARG(src && dst && n >= 1, "tnr_cast_f32_to_bf16: bad argument");
    hipLaunchKernelGGL(cast_f2b_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, n);
    TNR_CHECK_LAUNCH("tnr_cast_f32_to_bf16");
    return TNR_OK;
}
extern "C" int TNR_NAME(tnr_cast_bf16_to_f32)(const void* src, float* dst, int64_t n, void* stream) {
    TNR_CHECK_ARG(src && dst && n >= 1, "tnr_cast_bf16_to_f32: bad argument");
    hipLaunchKernelGGL(cast_b2f_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, dst, n);
    TNR_CHECK_LAUNCH("tnr_cast_bf16_to_f32");
    return TNR_OK;
}
