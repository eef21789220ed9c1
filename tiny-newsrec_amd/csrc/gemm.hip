// bf16 MFMA GEMMs for the encoder's Linear layers (forward, dgrad, wgrad).
//
//   tnr_gemm_nt       C[M,N] = epi(A[M,K] . B[N,K]^T)   both operands K-contiguous
//   tnr_gemm_tn_wgrad dW[N,K] = dY[M,N]^T . X[M,K]      both operands M-strided -> transposed LDS reads
//
// Tile 128x128, 4 waves (2x2), each wave 64x64 = 4x4 tiles of v_mfma_f32_16x16x32_bf16.
// Operands are staged global -> LDS with global_load_lds (16 B per lane, lane-linear LDS image); the
// XOR swizzle that makes the fragment reads bank-conflict free is applied on the per-lane SOURCE
// address and again on the read address (cdna_hip_programming.md section 5.4 rule 21).  Two LDS buffers,
// next tile's loads issued before the current tile's MFMAs, one barrier per K tile; 64 KB LDS per
// workgroup -> 2 workgroups per CU overlap each other's barrier stalls.
#include <algorithm>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace {

struct NTArgs {
    const bf16* A; int64_t lda;
    const bf16* B; int64_t ldb;
    void* C; int64_t ldc;
    int M, N, K;
    const float* bias;
    const bf16* res; int64_t ldres;
    bf16* aux; int64_t ldaux;
    int flags;
    float* colsum_part;        // TNR_EPI_COLSUM: (rows_of_partials, N) fp32, one row per 64-row strip of C
    int gm;                    // rasterisation group height
    int nt;                    // 1: streaming (non-temporal) accesses for once-touched epilogue operands
    int tile0;                 // first logical tile of this launch (0: one launch per GEMM)
    TnrDrop drop;              // TNR_EPI_DROPOUT: the site whose mask multiplies (acc + bias [-> activation]) before the residual add
    int mix_p, mix_x;          // ping-pong kernel: mix_p row panels, mix_x of them full height (32 MI rows), the others 32 rows
                               // shorter, spread evenly (pp_panel); mix_x == mix_p: the uniform tiling
    unsigned* queue;           // ping-pong kernel: 8 tile counters (one per XCD label) + a done counter, all zero between launches
    int probe;                 // unused by the product: the slot the probe builds' switches take (tools/probes/README.md); kept so that the
                               // kernel arguments of both builds have ONE layout (the kernels' scalar loads are scheduled around it)
    unsigned long long* clock; // ping-pong kernel, measurement hook (tnr_gemm_clock_stamps): workgroup b < clock_n writes the shader
    int clock_n;               // cycles (s_memtime) and 100 MHz ticks (s_memrealtime) of its life at clock[2 b], clock[2 b + 1]; NULL = off
};

constexpr int TILE_BYTES = 128 * 128;   // one operand tile: 128 rows x 64 bf16
constexpr int BUF_BYTES = 2 * TILE_BYTES;

// bijective XCD remap: blocks b and b+8 share an XCD (round-robin dispatch), give every XCD a
// contiguous run of the tile order so that neighbouring tiles (same A rows) hit one L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// Tile order inside the (XCD-contiguous) id space: groups of GM row-tiles x all column tiles, row-tile
// fastest, so the 32 workgroups an XCD runs at once form a GM x (32/GM) patch: each A row-tile is shared by
// 32/GM of them and each B column slice by GM (measured before: bn-fastest order, 71% L2 hit rate, every
// workgroup streaming its own B slice from beyond L2).
__device__ __forceinline__ void tile_coords(int wg, int nbm, int nbn, int GM, int& bm, int& bn) {
    const bool snake = GM < 0;           // odd row groups walk the column slices backwards (reuse the last B slices)
    GM = snake ? -GM : GM;
    int per = GM * nbn;
    int grp = wg / per;
    int first = grp * GM;
    int gsz = nbm - first < GM ? nbm - first : GM;
    int local = wg - grp * per;
    bm = first + local % gsz;
    bn = local / gsz;
    if (snake && (grp & 1)) bn = nbn - 1 - bn;
}

// erf-GELU / its derivative from an LDS table over [-8, 8), step 1/128 (2048 intervals), linear interpolation between
// exact node values.  The table holds the SATURATING factor -- Phi(x) for GELU(x) = x Phi(x), and GELU'(x) itself for
// the derivative -- so clamping the index is the whole range handling (Phi(-8) = 6e-16, Phi(8) = 1 - 6e-16) and an
// evaluation is fma, med3, fract, cvt, one 8-byte LDS gather, fma (+ mul): 7 VALU slots.  Interpolation error
// <= max|Phi''| h^2 / 8 = 1.8e-6 (x |x| <= 8 for GELU), the bf16 output rounding is 4e-3 relative.  The closed form cost
// ~35 slots per element and made the FFN epilogues VALU-bound (13 us per 256x256 tile).
constexpr int LUT_N = 2048;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// LDS accesses by 32-bit LDS address (the address of `smem` folded into the per-lane base once): `smem + offset` costs a v_add per
// access otherwise, and a vector instruction in a load segment waits ~16 clk for its issue slot
#define TNR_LDS(T, ADDR) (*(__attribute__((address_space(3))) T*)(size_t)(ADDR))
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
// node i holds {f(x_i), f(x_i+1) - f(x_i)}: ONE 8-byte gather per element on the 64-bank ds_read_b64 path
__device__ __forceinline__ void lut_build(f32x2* lut, bool grad) {
    for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) {
        float x0 = (float)(i - LUT_N / 2) * (1.0f / 128.0f), x1 = (float)(i + 1 - LUT_N / 2) * (1.0f / 128.0f);
        float e;
        float a = grad ? gelu_erf_grad(x0) : 0.5f * (1.0f + erf_as(x0 * 0.70710678118654752f, e));
        float b = grad ? gelu_erf_grad(x1) : 0.5f * (1.0f + erf_as(x1 * 0.70710678118654752f, e));
        lut[i] = (f32x2){a, b - a};
    }
}
template <bool GRAD>
__device__ __forceinline__ float lut_eval(const f32x2* lut, float x) {
    float t = __builtin_amdgcn_fmed3f(fmaf(x, 128.0f, (float)(LUT_N / 2)), 0.0f, (float)LUT_N - 0.001f);
    f32x2 e = lut[(int)t];
    float p = fmaf(__builtin_amdgcn_fractf(t), e[1], e[0]);
    return GRAD ? p : x * p;
}

// epilogue shared by the NT kernels: lane holds C[m_base + 16 i][n_base + 16 j + r], r = 0..3.
// All global reads of a 64-row strip (residual, aux) are issued before any arithmetic and the bias is loaded by
// the caller at kernel start: with the loads inside the (j, i) loops each one was a dependent L2 round trip
// (8 per tile, exposed), which cost more than the GELU arithmetic itself.
struct NTBias { f32x4 v[4]; };
__device__ __forceinline__ NTBias nt_load_bias(const NTArgs& g, int n_base) {
    NTBias b;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        b.v[j] = (g.flags & TNR_EPI_BIAS) ? *(const f32x4*)(g.bias + n_base + j * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
    return b;
}

__device__ __forceinline__ void nt_epilogue(const NTArgs& g, f32x4 (&acc)[4][4], int m_base, int n_base,
                                            const NTBias& bias, const f32x2* lut = nullptr) {
    const int flags = g.flags;
    bf16x4 rr[4][4], uu[4][4];
    if (flags & TNR_EPI_RES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = m_base + i * 16;
            m = m < g.M ? m : g.M - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[i][j] = *(const bf16x4*)(g.res + (int64_t)m * g.ldres + n_base + j * 16);
        }
    }
    if (flags & TNR_EPI_MULDGELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = m_base + i * 16;
            m = m < g.M ? m : g.M - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) uu[i][j] = *(const bf16x4*)(g.aux + (int64_t)m * g.ldaux + n_base + j * 16);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n_base + j * 16;
        f32x4 cs = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_base + i * 16;
            if (m >= g.M) continue;
            f32x4 v = acc[i][j] + bias.v[j];
            if (flags & TNR_EPI_AUXOUT) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
                *(bf16x4*)(g.aux + (int64_t)m * g.ldaux + n) = o;
            }
            if (flags & TNR_EPI_GELU) {
                if (lut) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = lut_eval<false>(lut, v[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
            }
            if (flags & TNR_EPI_TANH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = tnr_tanh(v[r]);
            }
            if (flags & TNR_EPI_MULDGELU) {
                if (lut) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= lut_eval<true>(lut, (float)uu[i][j][r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_grad((float)uu[i][j][r]);
                }
            }
            if (flags & TNR_EPI_DROPOUT) {      // BertSelfOutput / BertOutput: dense -> dropout -> + residual
                float dm[4];
                tnr_drop4(g.drop, (uint64_t)m * g.N + n, dm);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= dm[r];
            }
            if (flags & TNR_EPI_RES) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rr[i][j][r];
            }
            if (flags & TNR_EPI_OUTF32) {
                *(f32x4*)((float*)g.C + (int64_t)m * g.ldc + n) = v;
            } else {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = (bf16)v[r];
                    cs[r] += (float)o[r];
                }
                *(bf16x4*)((bf16*)g.C + (int64_t)m * g.ldc + n) = o;
            }
        }
        if (flags & TNR_EPI_COLSUM) {
            // column sums of this wave's 64-row strip: 16 lanes (lane & 15) hold the 16 rows of each tile
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = cs[r];
                t += __shfl_xor(t, 1, 64);
                t += __shfl_xor(t, 2, 64);
                t += __shfl_xor(t, 4, 64);
                t += __shfl_xor(t, 8, 64);
                cs[r] = t;
            }
            if ((threadIdx.x & 15) == 0) *(f32x4*)(g.colsum_part + (int64_t)(m_base >> 6) * g.N + n) = cs;
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NTArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int nbn = g.N >> 7;
    const int nbm = (g.M + 127) >> 7;
    const int wg = xcd_remap(blockIdx.x, nbm * nbn);
    int bm, bn;
    tile_coords(wg, nbm, nbn, 8, bm, bn);

    // ---- staging: 4 A pieces + 4 B pieces of 1 KiB (8 rows x 128 B) per wave and K tile
    const bf16* asrc[4];
    const bf16* bsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row = (w * 4 + q) * 8 + (lane >> 3);
        int chunk = (lane & 7) ^ (row & 7);
        int gm = bm * 128 + row;
        gm = gm < g.M ? gm : g.M - 1;                       // rows past M are computed but never stored
        asrc[q] = g.A + (int64_t)gm * g.lda + chunk * 8;
        bsrc[q] = g.B + (int64_t)(bn * 128 + row) * g.ldb + chunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * BUF_BYTES + (w * 4) * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(asrc[q] + kt * 64, base + q * 1024);
            glds16(bsrc[q] + kt * 64, base + TILE_BYTES + q * 1024);
        }
    };

    // fragment read offsets inside a tile: row = 16*t + (lane&15), 16-B chunk = 4*s + (lane>>4)
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = (lane & 15) * 128 + ((((4 * s) + (lane >> 4)) ^ (lane & 7)) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the same table GELU as the 256x256 kernel: results must not depend on which kernel a launch is routed to
    f32x2* lut = (f32x2*)(smem + 2 * BUF_BYTES);
    const bool use_lut = (g.flags & (TNR_EPI_GELU | TNR_EPI_MULDGELU)) != 0;
    if (use_lut) lut_build(lut, (g.flags & TNR_EPI_MULDGELU) != 0);
    const int nk = g.K >> 6;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sa = smem + cur * BUF_BYTES + (wm * 64) * 128;
        const char* sb = smem + cur * BUF_BYTES + TILE_BYTES + (wn * 64) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(sa + i * 16 * 128 + foff[s]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sb + j * 16 * 128 + foff[s]);
            // operands swapped: D = Bfrag . Afrag^T, so a lane holds C[m = lane&15][n = 4*(lane>>4) + r]
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = TNR_MFMA_16x16x32(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    nt_epilogue(g, acc, bm * 128 + wm * 64 + (lane & 15), bn * 128 + wn * 64 + (lane >> 4) * 4,
                nt_load_bias(g, bn * 128 + wn * 64 + (lane >> 4) * 4), use_lut ? lut : nullptr);
}

// ------------------------------------------------------------------------------------------------
// wgrad: out tile 128 (n) x 128 (k), reduction over m in steps of 64.  LDS tiles are [64 m][128 cols]
// bf16 (256-B rows); fragments come from ds_read_b64_tr_b16.  32-B column pairs are XOR-swizzled by
// (row&3) | ((row>>3)&1)<<2 so the 8 rows a 32-lane half touches cover all 64 banks once.
struct TNArgs {
    const bf16* dY; int64_t lddy;
    const bf16* X; int64_t ldx;
    float* ws;                 // (splits, N, K) fp32 slabs
    int Mt;                    // number of 64-row m tiles (Mpad / 64)
    int N, K;
    int tiles_per_split;
    int splits;
    unsigned* queue;           // ping-pong kernel: the stream's tile-queue counters (8 per-XCD-label unit counters + a done counter)
};

// Up to four weight gradients in ONE persistent launch (tnr_gemm_tn_wgrad_group): the (split, tile) units of the problems are
// numbered one after the other (ubase[i] = first unit of problem i; entries from n on hold the total) and pulled from the same
// queue.  Each unit is computed exactly as in the problem's own launch.
constexpr int TN_MAXP = 4;
struct TNGroup {
    TNArgs p[TN_MAXP];
    int ubase[TN_MAXP + 1];
    int xb[9];                 // unit range of XCD label x: [xb[x], xb[x + 1]) - equal shares of the WORK (units differ in length between problems)
    unsigned* queue;
};
// the eight ranges: one problem -> equal unit counts (the old rule); several -> cut where the cumulated m steps (+ a fixed cost per
// unit for its prologue and slab store) reach x / 8 of the total
static void tn_group_ranges(TNGroup& g, int n) {
    const int total = g.ubase[TN_MAXP];
    if (n == 1) {
        const int q8 = total >> 3, r8 = total & 7;
        for (int x = 0; x <= 8; ++x) g.xb[x] = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8;
        return;
    }
    int64_t w[TN_MAXP], cum[TN_MAXP + 1];
    cum[0] = 0;
    for (int i = 0; i < n; ++i) {
        w[i] = g.p[i].tiles_per_split + 8;
        cum[i + 1] = cum[i] + w[i] * (g.ubase[i + 1] - g.ubase[i]);
    }
    g.xb[0] = 0;
    g.xb[8] = total;
    for (int x = 1; x < 8; ++x) {
        const int64_t t = cum[n] * x / 8;
        int i = 0;
        while (i + 1 < n && cum[i + 1] <= t) ++i;
        int u = g.ubase[i] + (int)((t - cum[i] + w[i] / 2) / w[i]);
        if (u < g.xb[x - 1]) u = g.xb[x - 1];
        if (u > total) u = total;
        g.xb[x] = u;
    }
}

__device__ __forceinline__ int tn_swz(int row) { return (((row & 3) | (((row >> 3) & 1) << 2)) << 1); }

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TNArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wn = w >> 1, wk = w & 1;          // wave tile: 64 n x 64 k
    const int nbk = g.K >> 7;
    const int ntile = (g.N >> 7) * nbk;
    const int wg = xcd_remap(blockIdx.x, ntile * g.splits);   // same split (same rows of dY / X) on one XCD
    const int z = wg / ntile, tile = wg - z * ntile;
    const int bn = tile / nbk, bk = tile - bn * nbk;
    const int mt0 = z * g.tiles_per_split;
    int mt1 = mt0 + g.tiles_per_split;
    if (mt1 > g.Mt) mt1 = g.Mt;

    // staging: each tile = 64 rows x 256 B = 16 pieces of 1 KiB (4 rows); 4 pieces per wave per operand
    const bf16* ysrc[4];
    const bf16* xsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row = (w * 4 + q) * 4 + (lane >> 4);
        int chunk = (lane & 15) ^ tn_swz(row);
        ysrc[q] = g.dY + (int64_t)row * g.lddy + bn * 128 + chunk * 8;
        xsrc[q] = g.X + (int64_t)row * g.ldx + bk * 128 + chunk * 8;
    }
    auto stage = [&](int buf, int mt) {
        char* base = smem + buf * BUF_BYTES + (w * 4) * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(ysrc[q] + (int64_t)mt * 64 * g.lddy, base + q * 1024);
            glds16(xsrc[q] + (int64_t)mt * 64 * g.ldx, base + TILE_BYTES + q * 1024);
        }
    };

    // tr-read byte offsets: m row = 32*s + 8*(lane>>4) + q (+4), q = (lane&15)>>2, p = lane&3 ;
    // 16-col tile t -> logical 16-B chunk 2*t + (p>>1), byte (p&1)*8 inside it
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    int roff[2][2], rswz[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int row = 32 * s + 8 * g16 + q4 + 4 * h;
            roff[s][h] = row * 256 + (p4 & 1) * 8;
            rswz[s][h] = tn_swz(row);
        }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (mt0 < mt1) {
        stage(0, mt0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int mt = mt0; mt < mt1; ++mt) {
            const int cur = (mt - mt0) & 1;
            if (mt + 1 < mt1) stage(cur ^ 1, mt + 1);
            const char* sy = smem + cur * BUF_BYTES;
            const char* sx = smem + cur * BUF_BYTES + TILE_BYTES;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 yf[4], xf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    int cy = 2 * (wn * 4 + t) + (p4 >> 1);
                    int cx = 2 * (wk * 4 + t) + (p4 >> 1);
                    bf16x4 y0 = ds_read_tr16(sy + roff[s][0] + ((cy ^ rswz[s][0]) << 4));
                    bf16x4 y1 = ds_read_tr16(sy + roff[s][1] + ((cy ^ rswz[s][1]) << 4));
                    bf16x4 x0 = ds_read_tr16(sx + roff[s][0] + ((cx ^ rswz[s][0]) << 4));
                    bf16x4 x1 = ds_read_tr16(sx + roff[s][1] + ((cx ^ rswz[s][1]) << 4));
                    yf[t] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
                    xf[t] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                // D[row = k (regs)][col = n (lane&15)] : A operand = X^T fragment, B operand = dY fragment
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = TNR_MFMA_16x16x32(xf[j], yf[i], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // slab store: lane holds dW[n = .. + (lane&15)][k = .. + 4*(lane>>4) + r]
    float* slab = g.ws + (int64_t)z * g.N * g.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int n = bn * 128 + wn * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int k = bk * 128 + wk * 64 + j * 16 + (lane >> 4) * 4;
            *(f32x4*)(slab + (int64_t)n * g.K + k) = acc[i][j];
        }
    }
}


// ================================================================================================
// v2: 256x128 output tile, 8 waves (2 per SIMD, each 64x64 as above), 3-stage LDS ring of 48 KB stages
// ([sub-tile 0 | sub-tile 1 | other operand], 16 KB each).  Loads of K tile t+2 are issued while tile t
// is computed and stay in flight across the single barrier per K tile (counted s_waitcnt vmcnt, raw
// s_barrier).  Arithmetic intensity 85 FLOP per staged byte instead of 64: the 128x128 kernel above is
// bound by the ~70 GB/s one CU can pull from L2 (MI355X_MICROARCH.md, indexed-rows table).
constexpr int STAGE2 = 3 * TILE_BYTES;
constexpr int RING2 = 3 * STAGE2;

#define TNR_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(NTArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int nbn = g.N >> 7;
    const int nbm = (g.M + 255) >> 8;
    const int wg = xcd_remap(blockIdx.x, nbm * nbn);
    int bm, bn;
    tile_coords(wg, nbm, nbn, 4, bm, bn);

    // 48 pieces of 1 KiB per stage, 6 per wave: piece p -> sub-tile p>>4 (0,1 = A rows 0-127 / 128-255, 2 = B)
    const bf16* src[6];
    int dst[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        int p = w * 6 + q, sub = p >> 4, pp = p & 15;
        int row = pp * 8 + (lane >> 3);
        int chunk = (lane & 7) ^ (row & 7);
        if (sub < 2) {
            int gm = bm * 256 + sub * 128 + row;
            gm = gm < g.M ? gm : g.M - 1;
            src[q] = g.A + (int64_t)gm * g.lda + chunk * 8;
        } else {
            src[q] = g.B + (int64_t)(bn * 128 + row) * g.ldb + chunk * 8;
        }
        dst[q] = sub * TILE_BYTES + pp * 1024;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE2;
#pragma unroll
        for (int q = 0; q < 6; ++q) glds16(src[q] + kt * 64, base + dst[q]);
    };
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = (lane & 15) * 128 + ((((4 * s) + (lane >> 4)) ^ (lane & 7)) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = g.K >> 6;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    int cur = 0, nxt = 2;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) TNR_WAIT_VMCNT(6); else TNR_WAIT_VMCNT(0);     // tile kt landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                                    // ... and everybody else's; tile kt-1 fully read
        if (kt + 2 < nk) stage(nxt, kt + 2);
        const char* sa = smem + cur * STAGE2 + (wm >> 1) * TILE_BYTES + ((wm & 1) * 64) * 128;
        const char* sb = smem + cur * STAGE2 + 2 * TILE_BYTES + (wn * 64) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(sa + i * 16 * 128 + foff[s]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sb + j * 16 * 128 + foff[s]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = TNR_MFMA_16x16x32(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        cur = cur == 2 ? 0 : cur + 1;
        nxt = nxt == 2 ? 0 : nxt + 1;
    }
    nt_epilogue(g, acc, bm * 256 + wm * 64 + (lane & 15), bn * 128 + wn * 64 + (lane >> 4) * 4,
                nt_load_bias(g, bn * 128 + wn * 64 + (lane >> 4) * 4));
}

// wgrad v2: output tile 256 (n) x 128 (k); stage = [dY cols 0-127 | dY cols 128-255 | X], each [64 m][256 B]
__global__ __launch_bounds__(512, 2) void gemm_tn256_kernel(TNArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wn = w >> 1, wk = w & 1;          // wave tile: 64 n x 64 k
    const int nbk = g.K >> 7;
    const int ntile = (g.N >> 8) * nbk;
    const int wg = xcd_remap(blockIdx.x, ntile * g.splits);
    const int z = wg / ntile, tile = wg - z * ntile;
    const int bn = tile / nbk, bk = tile - bn * nbk;
    const int mt0 = z * g.tiles_per_split;
    int mt1 = mt0 + g.tiles_per_split;
    if (mt1 > g.Mt) mt1 = g.Mt;
    const int nt = mt1 - mt0;

    const bf16* src[6];
    int64_t ldsrc[6];
    int dst[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        int p = w * 6 + q, sub = p >> 4, pp = p & 15;
        int row = pp * 4 + (lane >> 4);
        int chunk = (lane & 15) ^ tn_swz(row);
        if (sub < 2) {
            src[q] = g.dY + (int64_t)(mt0 * 64 + row) * g.lddy + bn * 256 + sub * 128 + chunk * 8;
            ldsrc[q] = 64 * g.lddy;
        } else {
            src[q] = g.X + (int64_t)(mt0 * 64 + row) * g.ldx + bk * 128 + chunk * 8;
            ldsrc[q] = 64 * g.ldx;
        }
        dst[q] = sub * TILE_BYTES + pp * 1024;
    }
    auto stage = [&](int buf, int t) {
        char* base = smem + buf * STAGE2;
#pragma unroll
        for (int q = 0; q < 6; ++q) glds16(src[q] + (int64_t)t * ldsrc[q], base + dst[q]);
    };
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    int roff[2][2], rswz[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int row = 32 * s + 8 * g16 + q4 + 4 * h;
            roff[s][h] = row * 256 + (p4 & 1) * 8;
            rswz[s][h] = tn_swz(row);
        }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nt > 0) {
        stage(0, 0);
        if (nt > 1) stage(1, 1);
        int cur = 0, nxt = 2;
        for (int t = 0; t < nt; ++t) {
            if (t + 1 < nt) TNR_WAIT_VMCNT(6); else TNR_WAIT_VMCNT(0);
            __builtin_amdgcn_s_barrier();
            if (t + 2 < nt) stage(nxt, t + 2);
            const char* sy = smem + cur * STAGE2 + (wn >> 1) * TILE_BYTES;
            const char* sx = smem + cur * STAGE2 + 2 * TILE_BYTES;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 yf[4], xf[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    int cy = 2 * ((wn & 1) * 4 + tt) + (p4 >> 1);
                    int cx = 2 * (wk * 4 + tt) + (p4 >> 1);
                    bf16x4 y0 = ds_read_tr16(sy + roff[s][0] + ((cy ^ rswz[s][0]) << 4));
                    bf16x4 y1 = ds_read_tr16(sy + roff[s][1] + ((cy ^ rswz[s][1]) << 4));
                    bf16x4 x0 = ds_read_tr16(sx + roff[s][0] + ((cx ^ rswz[s][0]) << 4));
                    bf16x4 x1 = ds_read_tr16(sx + roff[s][1] + ((cx ^ rswz[s][1]) << 4));
                    yf[tt] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
                    xf[tt] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = TNR_MFMA_16x16x32(xf[j], yf[i], acc[i][j], 0, 0, 0);
            }
            cur = cur == 2 ? 0 : cur + 1;
            nxt = nxt == 2 ? 0 : nxt + 1;
        }
    }
    float* slab = g.ws + (int64_t)z * g.N * g.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int n = bn * 256 + wn * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int k = bk * 128 + wk * 64 + j * 16 + (lane >> 4) * 4;
            *(f32x4*)(slab + (int64_t)n * g.K + k) = acc[i][j];
        }
    }
}

// ---- row-contiguous epilogue of the 256x256 kernels -------------------------------------------------
// The accumulator layout gives a lane 4 consecutive n of one row: 8-byte bf16 accesses, 16 rows per wave
// instruction.  Measured: an extra bf16 tensor written/read that way (AUXOUT / RES / MULDGELU) cost +220..290 us
// per 52800x3072 launch while an fp32 output (16-byte accesses, twice the bytes) cost +40 us -- the epilogue is
// bound by the number of small memory transactions.  So the fp32 tile goes through LDS in two passes of 128 rows
// ([128][256] fp32, rows padded by 16 B) and is post-processed with thread -> (row, 8 consecutive columns):
// 16-byte residual / aux loads and C / aux stores, two full 512-byte row segments per wave instruction.
constexpr int EPI_LD = 256 * 4 + 16;                 // bytes per staged row
constexpr int EPI_BYTES = 128 * EPI_LD;              // 133,120 B
constexpr int LDS3_BYTES = EPI_BYTES + LUT_N * 8;

// Optional streaming (non-temporal) accesses for the epilogue's once-touched operands (C / side-output stores, residual /
// pre-activation loads), env TNR_GEMM_NT=1.  Measured with an interleaved same-box A/B over seven shapes: within +-0.8 %
// of plain accesses (an earlier "+7 %" was box-to-box variance), so the default is plain.
#ifndef TNR_GEMM_NT
#define TNR_GEMM_NT 1
#endif
#if TNR_GEMM_NT
#define TNR_NT_STORE(v, p) do { if (g.nt) __builtin_nontemporal_store((v), (p)); else *(p) = (v); } while (0)
#define TNR_NT_LOAD(p) (g.nt ? __builtin_nontemporal_load((p)) : *(p))
#else
#define TNR_NT_STORE(v, p) (*(p) = (v))
#define TNR_NT_LOAD(p) (*(p))
#endif

// MI x NJ 16x16 accumulator blocks per wave, NTHR threads per workgroup (8 waves of 128x64 or 4 waves of 128x128)
template <int MI, int NJ = 4, int NTHR = 512>
__device__ __forceinline__ void nt_epilogue_coalesced(const NTArgs& g, f32x4 (&acc)[MI][NJ], char* smem, const f32x2* lut,
                                                      int bm, int bn, int wm, int wn, int lane) {
    constexpr int PR = 16 * MI;          // rows per pass (= rows per wave), tile height 2 * PR
    constexpr int RG = NTHR / 32;        // row groups working on a pass
    constexpr int ITER = PR / RG;        // rows per thread and pass
    const int flags = g.flags;
    const int tid = threadIdx.x;
    const int c8 = (tid & 31) * 8, rg = tid >> 5;            // 8 columns, row group 0..15
    const int n = bn * 256 + c8;
    f32x4 b0 = (f32x4){0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (flags & TNR_EPI_BIAS) {
        b0 = *(const f32x4*)(g.bias + n);
        b1 = *(const f32x4*)(g.bias + n + 4);
    }
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    // The residual / pre-activation rows are HBM reads the arithmetic below waits for (an extra 16-bit M x N operand
    // cost +50 us per 52800x3072 launch with the loads issued where they are used): pass 0's are issued before the
    // accumulators are staged, pass 1's while pass 0 is being processed.
    // One prefetched operand per launch (the engine never combines RES with MULDGELU; if both are set the residual
    // is read where it is used).
    const bool pre_aux = (flags & TNR_EPI_MULDGELU) != 0;
    const bool pre_res = !pre_aux && (flags & TNR_EPI_RES) != 0;
    bf16x8 xx[2][ITER];
    auto issue_loads = [&](int pass) {
        const int m0 = bm * (2 * PR) + pass * PR;
        if (pre_aux | pre_res) {
            const bf16* src = pre_aux ? g.aux : g.res;
            const int64_t ld = pre_aux ? g.ldaux : g.ldres;
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                int m = m0 + rg + RG * it;
                m = m < g.M ? m : g.M - 1;
                xx[pass][it] = TNR_NT_LOAD((const bf16x8*)(src + (int64_t)m * ld + n));
            }
        }
    };
    issue_loads(0);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();                                       // staging area free (K loop / previous pass done)
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    *(f32x4*)(smem + (i * 16 + (lane & 15)) * EPI_LD + (wn * (NJ * 16) + j * 16 + (lane >> 4) * 4) * 4) = acc[i][j];
        }
        __syncthreads();
        const int m0 = bm * (2 * PR) + pass * PR;
        if (pass == 0) issue_loads(1);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int row = rg + RG * it;
            const int m = m0 + row;
            if (m >= g.M) continue;
            f32x4 v0 = *(const f32x4*)(smem + row * EPI_LD + c8 * 4) + b0;
            f32x4 v1 = *(const f32x4*)(smem + row * EPI_LD + c8 * 4 + 16) + b1;
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (flags & TNR_EPI_AUXOUT) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                TNR_NT_STORE(o, (bf16x8*)(g.aux + (int64_t)m * g.ldaux + n));
            }
            if (flags & TNR_EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = lut_eval<false>(lut, v[e]);
            }
            if (flags & TNR_EPI_TANH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tnr_tanh(v[e]);
            }
            if (flags & TNR_EPI_MULDGELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= lut_eval<true>(lut, (float)xx[pass][it][e]);
            }
            if (flags & TNR_EPI_DROPOUT) {
                float dm[8];
                tnr_drop8(g.drop, ((uint64_t)m * g.N + n) >> 3, dm);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dm[e];
            }
            if (flags & TNR_EPI_RES) {
                bf16x8 r = pre_res ? xx[pass][it] : *(const bf16x8*)(g.res + (int64_t)m * g.ldres + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
            }
            if (flags & TNR_EPI_OUTF32) {
                float* c = (float*)g.C + (int64_t)m * g.ldc + n;
                TNR_NT_STORE(((f32x4){v[0], v[1], v[2], v[3]}), (f32x4*)c);
                TNR_NT_STORE(((f32x4){v[4], v[5], v[6], v[7]}), (f32x4*)(c + 4));
            } else {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = (bf16)v[e];
                    cs[e] += (float)o[e];
                }
                TNR_NT_STORE(o, (bf16x8*)((bf16*)g.C + (int64_t)m * g.ldc + n));
            }
        }
    }
    if (flags & TNR_EPI_COLSUM) {
        // column sums of the 256-row tile: 16 row groups hold partials for the same 8 columns
        __syncthreads();
        float* red = (float*)smem;                             // [RG][256]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rg * 256 + c8 + e] = cs[e];
        __syncthreads();
        if (tid < 256) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RG; ++r) t += red[r * 256 + tid];
            float* pr = g.colsum_part + (int64_t)(bm * 4) * g.N + bn * 256 + tid;   // 4 partial rows per row-tile (v2 layout)
            pr[0] = t;
            pr[g.N] = 0.f;
            pr[2 * (int64_t)g.N] = 0.f;
            pr[3 * (int64_t)g.N] = 0.f;
        }
    }
}

// ================================================================================================
// v3: 256x256 output tile, 8 waves (2 x 4, each 128 x 64), two 64 KB LDS stages
// ([A rows 0-127 | A rows 128-255 | B rows 0-127 | B rows 128-255], 16 KB each).  128 FLOP per staged byte:
// measured on MI355X the v2 load pipeline alone (no MFMA) took 80% of the kernel time, i.e. these GEMMs are
// bound by the ~10-12 TB/s the CUs can stream from L2 into LDS, so the tile has to grow, not the schedule.
constexpr int STAGE3 = 4 * TILE_BYTES;
constexpr int RING3 = 2 * STAGE3;

// MI = 16-row MFMA tiles per wave along M: tile height BM = 32 * MI (256 or 224).  The host picks the height
// that minimises ceil(tiles / CUs) * BM for the launch (e.g. N = 768: 621 tiles of 256 rows = 3 rounds on 256 CUs;
// 708 tiles of 224 rows are 3 rounds too, each 12.5 % shorter).
template <int MI>
__global__ __launch_bounds__(512, 2) void gemm_nt256x256_kernel(NTArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PR = 16 * MI, BM = 2 * PR;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int nbn = g.N >> 8;
    const int nbm = (g.M + BM - 1) / BM;
    const int wg = g.tile0 + xcd_remap(blockIdx.x, gridDim.x);      // gridDim.x = tiles of this launch
    int bm, bn;
    tile_coords(wg, nbm, nbn, g.gm, bm, bn);

    const bf16* src[8];
    int dst[8];
    bool live[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int p = w * 8 + q, sub = p >> 4, pp = p & 15;
        int row = pp * 8 + (lane >> 3);
        int chunk = (lane & 7) ^ (row & 7);
        live[q] = sub >= 2 || pp * 8 < PR;                 // A sub-tiles hold PR rows (PR/8 pieces of 8 rows)
        if (sub < 2) {
            int gm = bm * BM + sub * PR + row;
            gm = gm < g.M ? gm : g.M - 1;
            src[q] = g.A + (int64_t)gm * g.lda + chunk * 8;
        } else {
            src[q] = g.B + (int64_t)(bn * 256 + (sub - 2) * 128 + row) * g.ldb + chunk * 8;
        }
        dst[q] = sub * TILE_BYTES + pp * 1024;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE3;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (MI == 8 || live[q]) glds16(src[q] + kt * 64, base + dst[q]);     // wave-uniform predicate
    };
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = (lane & 15) * 128 + ((((4 * s) + (lane >> 4)) ^ (lane & 7)) << 4);

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x2* lut = (f32x2*)(smem + EPI_BYTES);             // own LDS region, built while the first loads fly
    if (g.flags & (TNR_EPI_GELU | TNR_EPI_MULDGELU)) lut_build(lut, (g.flags & TNR_EPI_MULDGELU) != 0);
    const int nk = g.K >> 6;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        TNR_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();            // tile kt landed everywhere; tile kt-1 fully consumed
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sa = smem + cur * STAGE3 + wm * TILE_BYTES;
        const char* sb = smem + cur * STAGE3 + (2 + (wn >> 1)) * TILE_BYTES + ((wn & 1) * 64) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[MI], bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sb + j * 16 * 128 + foff[s]);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(sa + i * 16 * 128 + foff[s]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = TNR_MFMA_16x16x32(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    nt_epilogue_coalesced<MI>(g, acc, smem, lut, bm, bn, wm, wn, lane);
}

// ---- epilogue of the ping-pong kernel: no transposes, no LDS staging ----------------------------------
// The MFMA is called with the B fragment first, so lane (m = lane & 15, q = lane >> 4) holds, in register r of block j,
// the output of row m and of the B row that lane 4 q + r fed into block j.  WHICH B row that is is free: the kernel's
// fragment reads take row 32 (j >> 1) + 8 q' + 4 (j & 1) + r' for lane q' r' (pp_brow), so lane q ends up with columns
// 8 q .. 8 q + 7 (blocks 0, 1) and 32 + 8 q .. 32 + 8 q + 7 (blocks 2, 3) of the wave's 64: two 16-byte stores per row and
// lane, four lanes writing 64 contiguous bytes of a row per instruction.  (The first version of this kernel kept the
// natural row order and moved the data with 128 v_permlane swaps per wave and tile; those and the runtime flag tests
// made the epilogue 8 of the 30 us of a K = 768 tile, tools/gemm_probe.py.)  CF >= 0: the epilogue flags at compile time
// (one instance per combination the engine uses), CF < 0: runtime flags (anything else).
// Bias comes from LDS (staged by the prologue's LDS-DMA: an ordinary global load here would make the compiler wait
// vmcnt(0), i.e. for the next tile's first operand loads); residual / pre-activation operands are read in the same
// layout, prefetched one block ahead; column sums (bias gradient of the producing Linear) are reduced over the 16 rows
// of a block by row-wise shuffles and over blocks in registers.
// Full-line stores.  Lane (m = lane & 15, q = lane >> 4) holds two 16-byte pieces of ONE output row: columns 8 q .. 8 q + 7 and
// 32 + 8 q .. of the wave's 64.  Stored as they stand, one instruction writes 16 rows x 64 bytes and the next the other halves
// of the same 128-byte lines: 77-80 cycles per store instruction and CU, against 45-59 when ADJACENT lanes write the two halves of
// a line in one instruction (tools/scratch/store_path.hip: lane pairs (m, m ^ 1) reach the rate of fully contiguous stores, pairs
// (m, m ^ 4) or (m, m ^ 8) do not - the coalescer looks at neighbouring lanes; the store path is what the epilogue costs, DESIGN.md
// section 4 item 15).  So the lanes of rows m and m ^ 1 trade a piece first: afterwards `u` is a piece of the pair's EVEN row and
// `v` one of its ODD row, at columns 8 q (even lane) or 32 + 8 q (odd lane) - each instruction then writes 8 rows x 128 bytes.
// One v_cndmask_b32_dpp per register (quad_perm [1, 0, 3, 2] on the first source, lane parity in VCC); the s_nop covers the
// VALU-write -> DPP-read hazard the compiler does not pad inside an asm statement.
__device__ __forceinline__ void pp_pair_exchange(const bf16x8& o0, const bf16x8& o1, u32x4& u, u32x4& v) {
    const u32x4 a = __builtin_bit_cast(u32x4, o0), b = __builtin_bit_cast(u32x4, o1);
    unsigned u0, u1, u2, u3, v0, v1, v2, v3;
    // u = even lane ? own first piece : the even partner's second piece
    asm("s_mov_b64 vcc, %12\n\ts_nop 1\n\t"
        "v_cndmask_b32_dpp %0, %4, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %5, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %2, %6, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %3, %7, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(0x5555555555555555ull)
        : "vcc");
    // v = odd lane ? own second piece : the odd partner's first piece
    asm("s_mov_b64 vcc, %12\n\ts_nop 1\n\t"
        "v_cndmask_b32_dpp %0, %4, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %5, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %2, %6, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %3, %7, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "s"(0xAAAAAAAAAAAAAAAAull)
        : "vcc");
    u = (u32x4){u0, u1, u2, u3};
    v = (u32x4){v0, v1, v2, v3};
}

template <int MI, int CF>
// The tile is row panel bm, rows rs .. rs + 32 nblk - 1: nblk (MI, or MI - 1 for a short tile) 16-row blocks per wave group.
__device__ __forceinline__ void nt_epilogue_cols(const NTArgs& g, f32x4 (&acc)[MI][4], const f32x2* lut, const float* bias_lds,
                                                 int bm, int bn, int rs, int nblk, int wm, int wn, int lane) {
    const int flags = CF >= 0 ? CF : g.flags;
    const int m16 = lane & 15, qd = lane >> 4;
    const int n0 = bn * 256 + wn * 64 + qd * 8;                  // this lane's columns: n0 .. n0 + 7 and n0 + 32 .. n0 + 39
    const int row0 = rs + wm * 16 * nblk + m16;
    const bool odd = (m16 & 1) != 0;
    const int rowp = row0 - (m16 & 1), n0p = n0 + (odd ? 32 : 0);   // full-line stores: the pair's even row, this lane's column piece
    constexpr bool do_lut = true, do_xload = true;
    const int n0s = n0, row0s = row0;
    const int n0ps = n0p, rowps = rowp;
    constexpr bool do_store = true;
    (void)n0s; (void)row0s;
    float bb[16];
    if (flags & TNR_EPI_BIAS) {
        const float* bl = bias_lds + wn * 64 + qd * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
                f32x4 t = *(const f32x4*)(bl + 32 * h + e);
                bb[8 * h + e] = t[0]; bb[8 * h + e + 1] = t[1]; bb[8 * h + e + 2] = t[2]; bb[8 * h + e + 3] = t[3];
            }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) bb[e] = 0.f;
    }
    float cs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[e] = 0.f;
    const bool pre_aux = (flags & TNR_EPI_MULDGELU) != 0;
    const bool pre_res = !pre_aux && (flags & TNR_EPI_RES) != 0;
    // one prefetched 16-bit operand row segment per block (the engine never combines RES with MULDGELU)
    const bf16* xsrc = pre_aux ? (const bf16*)g.aux : g.res;
    const int64_t xld = pre_aux ? g.ldaux : g.ldres;
    bf16x8 x0 = {}, x1 = {};
    auto xload = [&](int i, bf16x8& a, bf16x8& b) {
        if (!do_xload) return;
        int m = row0 + i * 16;
        m = m < g.M ? m : g.M - 1;
        const bf16* p = xsrc + (int64_t)m * xld + n0;
        a = *(const bf16x8*)p;
        b = *(const bf16x8*)(p + 32);
    };
    if (pre_aux | pre_res) xload(0, x0, x1);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (i == MI - 1 && nblk < MI) break;                     // short tile
        bf16x8 y0 = x0, y1 = x1;
        if ((pre_aux | pre_res) && i + 1 < MI && i + 1 < nblk) xload(i + 1, x0, x1);
        const int m = row0 + i * 16;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 t = acc[i][j];
            v[4 * j] = t[0] + bb[4 * j]; v[4 * j + 1] = t[1] + bb[4 * j + 1];
            v[4 * j + 2] = t[2] + bb[4 * j + 2]; v[4 * j + 3] = t[3] + bb[4 * j + 3];
        }
        const bool live = m < g.M;
        if (flags & TNR_EPI_AUXOUT) {
            bf16x8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) { o0[e] = (bf16)v[e]; o1[e] = (bf16)v[8 + e]; }
            u32x4 pu, pv;
            pp_pair_exchange(o0, o1, pu, pv);
            if (do_store) {
                bf16* p = g.aux + (int64_t)(rowps + i * 16) * g.ldaux + n0ps;
                if (rowp + i * 16 < g.M) *(u32x4*)p = pu;
                if (rowp + i * 16 + 1 < g.M) *(u32x4*)(p + g.ldaux) = pv;
            }
        }
        if (flags & TNR_EPI_GELU) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = do_lut ? lut_eval<false>(lut, v[e]) : v[e] * 0.5f;
        }
        if (flags & TNR_EPI_TANH) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = tnr_tanh(v[e]);
        }
        if (flags & TNR_EPI_MULDGELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] *= do_lut ? lut_eval<true>(lut, (float)y0[e]) : (float)y0[e];
                v[8 + e] *= do_lut ? lut_eval<true>(lut, (float)y1[e]) : (float)y1[e];
            }
        }
        if (flags & TNR_EPI_DROPOUT) {
            float dm[8];
            tnr_drop8(g.drop, ((uint64_t)m * g.N + n0) >> 3, dm);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= dm[e];
            tnr_drop8(g.drop, ((uint64_t)m * g.N + n0 + 32) >> 3, dm);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[8 + e] *= dm[e];
        }
        if (flags & TNR_EPI_RES) {
            bf16x8 r0 = y0, r1 = y1;
            if (pre_aux) {                 // both operands in one launch: the residual is read where it is used
                const bf16* p = g.res + (int64_t)(live ? m : g.M - 1) * g.ldres + n0;
                r0 = *(const bf16x8*)p;
                r1 = *(const bf16x8*)(p + 32);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] += (float)r0[e];
                v[8 + e] += (float)r1[e];
            }
        }
        if (flags & TNR_EPI_OUTF32) {
            if (live && do_store) {
                float* c = (float*)g.C + (int64_t)(row0s + i * 16) * g.ldc + n0s;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 8; e += 4)
                        *(f32x4*)(c + 32 * h + e) = (f32x4){v[8 * h + e], v[8 * h + e + 1], v[8 * h + e + 2], v[8 * h + e + 3]};
            }
        } else {
            bf16x8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) { o0[e] = (bf16)v[e]; o1[e] = (bf16)v[8 + e]; }
            u32x4 pu, pv;
            pp_pair_exchange(o0, o1, pu, pv);
            if (do_store) {
                bf16* c = (bf16*)g.C + (int64_t)(rowps + i * 16) * g.ldc + n0ps;
                if (rowp + i * 16 < g.M) *(u32x4*)c = pu;
                if (rowp + i * 16 + 1 < g.M) *(u32x4*)(c + g.ldc) = pv;
            }
            if ((flags & TNR_EPI_COLSUM) && live) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { cs[e] += (float)o0[e]; cs[8 + e] += (float)o1[e]; }
            }
        }
    }
    if (flags & TNR_EPI_COLSUM) {
        // column sums over this wave's 16 MI rows: 16 lanes (one row of the wave) hold the same 16 columns
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float t = cs[e];
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 4, 64);
            t += __shfl_xor(t, 8, 64);
            cs[e] = t;
        }
        if (m16 == 0) {
            // four partial rows per row tile (tnr_gemm_colsum_rows): rows 0, 1 = the two row halves, rows 2, 3 = 0
            float* pr = g.colsum_part + ((int64_t)(bm * 4) + wm) * g.N + n0;
            float* pz = g.colsum_part + ((int64_t)(bm * 4) + 2 + wm) * g.N + n0;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 8; e += 4) {
                    *(f32x4*)(pr + 32 * h + e) = (f32x4){cs[8 * h + e], cs[8 * h + e + 1], cs[8 * h + e + 2], cs[8 * h + e + 3]};
                    *(f32x4*)(pz + 32 * h + e) = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
        }
    }
}

// ================================================================================================
// v8 "ping-pong": the v3 tile (256 x 256 x 64, 8 waves of 128 x 64, two 64 KB K stages) with the two wave groups of a
// SIMD staggered by one barrier.  Waves w and w + 4 share a SIMD; group 0 (w < 4, rows 0-127) and group 1 (rows 128-255)
// alternate between a LOAD segment (fragment ds_reads + the LDS-DMA of a later half tile) and an MFMA segment (16 MFMAs:
// one 64 x 32 quadrant of the wave's 128 x 64 over K = 64), so that while one wave of a SIMD issues MFMAs its partner's
// memory instructions run beside them:
//      interval      0    1    2    3    4    5    6    7   | 8 ...
//      group 0       L0   M0   L1   M1   L2   M2   L3   M3  | L0' ...
//      group 1       -    L0   M0   L1   M1   L2   M2   L3  | M3  L0' ...
// K tile = 4 phases: L0 reads A rows 0-63, L1 B cols 32-63, L2 A rows 64-127, L3 B cols 0-31 OF THE NEXT K TILE (B is kept in
// registers across the K tile; the two B register pairs swap roles every K tile, see TNR_PP_KTILE);
// M0..M3 = quadrants (A lo, B lo), (A lo, B hi), (A hi, B hi), (A hi, B lo).
// The LDS-DMA stream (half tiles of 16 KB: A0 = rows 0-127, A1 = rows 128-255, B0, B1): phase p of K tile t issues
// B0(t+1) + B1(t+1), A1(t+1), nothing, A0(t+2).  Hazards:
//   WAR  a half tile is re-staged only after its last reader's lgkmcnt(0) AND a barrier both groups have passed
//        (A0(t): group 0's L2 -> free from interval 8t+6 = this phase 3; A1(t-1): group 1's L2 of tile t-1; B(t-1): L1 of tile
//        t-1 and, for its columns 0-31, L3 of tile t-2);
//   RAW  phase 2 waits vmcnt(2) (everything but the A1(t+1) pieces: B(t+1) has landed) and both groups pass a barrier before
//        phase 3 reads B(t+1); phase 3 waits vmcnt(2) (everything but the A0(t+2) pieces) and both groups pass a barrier
//        before tile t+1's first A read.  The streamed operand A gets 4+ intervals, the L2-resident weights B 4.
// All waves execute the same number of barriers (group 1 one extra before the loop, group 0 one extra after it).
// B rows are staged with their own swizzle (pp_bswz) and read in the permuted order of nt_epilogue_cols:
//   fragment row of lane n' = lane & 15 in block j:  32 (j >> 1) + 8 (n' >> 2) + 4 (j & 1) + (n' & 3)
//   16-byte slot of (row, chunk):  chunk ^ pp_bswz(row),  pp_bswz(row) = row bit 1 | row bits 3-4 << 1
// which keeps every 16-lane group of a ds_read_b128 on 16 distinct slots of the 256-byte bank row (the rows of a group
// differ in bits 0-1 and 3-4; bit 0 selects the 128-byte half by itself, the other three go through the XOR).
// Row panels of two heights.  A launch's time is (tiles per workgroup, rounded UP) x (time per tile), and with one height the
// rounding costs 3-9 % on the step's shapes (M = 52 800: N = 2304 is 8.3 rounds of 224-row tiles, N = 768 2.77).  With P panels
// of which x are BMAX rows high and the rest BMAX - 32, P x (N / 256) can be made a multiple of the workgroup count and the
// rows still add up to M; the kernel skips the last 16-row block of both wave groups in a short tile.  The tall panels are
// spread evenly over the P (every XCD's run of the tile order gets the same mix; the queue evens out the rest):
// panel p starts at row (BMAX - 32) p + 32 floor(p x / P) and is tall iff floor((p + 1) x / P) > floor(p x / P).
__device__ __forceinline__ void pp_panel(int p, int P, int x, int bmax, int& row0, bool& tall) {
    const int a = (p * x) / P, b = ((p + 1) * x) / P;
    row0 = p * (bmax - 32) + 32 * a;
    tall = b > a;
}
__device__ __forceinline__ int pp_bswz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

// Tile queue of the ping-pong kernel.  pp_q_fetch issues `old = (*ctr)++` from lane 0 of the calling wave when `on` (wave-
// uniform) and does NOT wait: the compiler is not told about the operation (it would fold the single-lane atomic into a wave
// reduction whose readfirstlane needs the result at once, and wait for it right behind the tile-start barrier), so `dst` is
// valid only behind pp_q_wait, which every consumer goes through.  No branch around either: a phi copy of `dst` ahead of the
// wait would read the register before the data is there.
// (a counter set: 8 tile counters + the count of workgroups that have left, each on a 256-byte line of its own - the workgroups
// of one XCD label then update a line that stays in their L2 instead of passing it between the eight)
constexpr int PP_Q_STRIDE = 64;
[[maybe_unused]] constexpr int PP_Q_SET = 9 * PP_Q_STRIDE;
__device__ __forceinline__ void pp_q_fetch(unsigned& dst, unsigned* ctr, bool on) {
    const unsigned long long m = (unsigned)__builtin_amdgcn_readfirstlane(on ? 1 : 0);
    unsigned long long sv;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %5\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                 : "=&v"(dst), "=&s"(sv) : "v"(0u), "v"(1u), "s"(ctr), "s"(m) : "memory");
}
__device__ __forceinline__ void pp_q_wait(unsigned& v) { asm volatile("s_waitcnt vmcnt(0) ; tile queue: %0" : "+v"(v) :: "memory"); }

template <int N_> __device__ __forceinline__ void tnr_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int MI, int CF>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(NTArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PR = 16 * MI, BM = 2 * PR;
    const int flags = CF >= 0 ? CF : g.flags;
    constexpr int APIECES = PR / 8;                         // 1 KiB pieces (8 rows) per A half tile: 16 or 14
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the stagger branches are scalar
    const int wm = w >> 2, wn = w & 3;
    const int nbn = g.N >> 8;
    const int nbm = g.mix_p;
    // Persistent: one workgroup per CU; the workgroups of an XCD label (blocks b and b + 8 share an XCD) PULL consecutive tiles
    // of that label's contiguous run of the tile order from a counter, so the 32 of them work on a GM x (32 / GM) patch that
    // shares A / B panels through the XCD's L2, as a plain launch of one workgroup per tile would - and a workgroup that starts
    // late or not at all (CUs held by a collective on another stream: with static shares 8 held CUs cost 28 %, tools/
    // cu_contention.py) only shifts work to the others.  The index of the next tile is fetched at the start of a tile's K loop
    // and read at its end (in time for the next tile's first loads to go out ahead of the epilogue), so only the first fetch's
    // latency shows; no deeper lookahead - a tile claimed early is a tile another workgroup cannot take (with two claims
    // up front, a launch of one tile per workgroup ran two tiles on half of them).  The last workgroup to leave resets the
    // counters.
    const int ntile = nbm * nbn;
    const int xcd = blockIdx.x & 7;
    const int q8 = ntile >> 3, r8 = ntile & 7;
    const int c0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int c1 = c0 + (xcd < r8 ? q8 + 1 : q8);
    int* const qlds = (int*)(smem + LDS3_BYTES);
    unsigned q0;                                         // the first tile: in flight while the tables below are set up
    pp_q_fetch(q0, g.queue + xcd * PP_Q_STRIDE, w == 0);
    auto leave = [&]() {                                 // last workgroup out zeroes the counters for the next launch
        if (tid == 0) {
            const unsigned d = __hip_atomic_fetch_add(g.queue + 8 * PP_Q_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == gridDim.x - 1)
                for (int i = 0; i < 9; ++i)
                    __hip_atomic_store(g.queue + i * PP_Q_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    unsigned long long pt0 = 0, pr0 = 0;                 // tnr_gemm_clock_stamps: the clock the chip holds under THIS kernel
    if (g.clock) { pt0 = __builtin_amdgcn_s_memtime(); pr0 = __builtin_amdgcn_s_memrealtime(); }

    // staging: every wave issues pieces 2w, 2w+1 of every half tile (wave 7 has no A pieces in the 224-row variant)
    const bf16* srcA[2][2];
    const bf16* srcB[2][2];
    const bool a_live = MI == 8 || 2 * w + 1 < APIECES;
    auto set_src = [&](int t) {
        int bm_, bn_, rs_;
        bool tall_;
        tile_coords(t, nbm, nbn, g.gm, bm_, bn_);
        pp_panel(bm_, nbm, g.mix_x, BM, rs_, tall_);
        const int pr_ = tall_ ? PR : PR - 16;            // rows per wave group; a short tile's last pieces re-read its last row
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = (2 * w + q) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ (row & 7), chunkb = (lane & 7) ^ pp_bswz(row);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int gm = rs_ + h * pr_ + (row < pr_ ? row : pr_ - 1);
                gm = gm < g.M ? gm : g.M - 1;
                srcA[h][q] = g.A + (int64_t)gm * g.lda + chunk * 8;
                srcB[h][q] = g.B + (int64_t)(bn_ * 256 + h * 128 + row) * g.ldb + chunkb * 8;
            }
        }
        return bn_;
    };
    // half tile `which` (0 A0, 1 A1, 2 B0, 3 B1) of K tile kt into stage buffer kt & 1
    auto issue = [&](int which, int kt) {
        char* base = smem + (kt & 1) * STAGE3 + which * TILE_BYTES + (2 * w) * 1024;
        if (which < 2) {
            if (a_live) {
                glds16(srcA[which][0] + kt * 64, base);
                glds16(srcA[which][1] + kt * 64, base + 1024);
            }
        } else {
            glds16(srcB[which - 2][0] + kt * 64, base);
            glds16(srcB[which - 2][1] + kt * 64, base + 1024);
        }
    };
    const int nk = g.K >> 6;
    // bias of the tile's 256 columns: 1 KiB by one LDS-DMA of wave 0 into one of two buffers behind the stage ring (tile
    // parity), oldest operation of the prologue so that every later counted wait covers it
    float* const bias_lds = (float*)(smem + RING3);
    auto prologue = [&](int t, int par) {                // the first five half tiles of tile t's K loop
        const int bn_ = set_src(t);
        if ((flags & TNR_EPI_BIAS) && w == 0) glds16(g.bias + bn_ * 256 + lane * 4, bias_lds + par * 256);
        issue(0, 0);
        issue(1, 0);
        issue(2, 0);
        issue(3, 0);
        if (nk > 1) issue(0, 1);
    };
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = (lane & 15) * 128 + ((((4 * s) + (lane >> 4)) ^ (lane & 7)) << 4);
    int boff[2];
    {
        const int rb = 8 * ((lane & 15) >> 2) + (lane & 3);      // + 32 (j >> 1) + 4 (j & 1): bits the swizzle does not use
#pragma unroll
        for (int s = 0; s < 2; ++s) boff[s] = rb * 128 + ((((4 * s) + (lane >> 4)) ^ pp_bswz(rb)) << 4);
    }

    f32x2* lut = (f32x2*)(smem + EPI_BYTES);             // own LDS region, built while the queue answers
    if (flags & (TNR_EPI_GELU | TNR_EPI_MULDGELU)) lut_build(lut, (flags & TNR_EPI_MULDGELU) != 0);
    pp_q_wait(q0);
    if (tid == 0) qlds[0] = c0 + (int)q0;
    __syncthreads();
    int tile = qlds[0];
    if (tile >= c1) { leave(); return; }                 // whole workgroup, before any other barrier
    int par = 0;
    prologue(tile, par);

    constexpr int ILO = MI < 4 ? MI : 4, IHI = MI - ILO; // 16-row blocks of the wave's lower / upper A half
    bf16x8 af[4][2], bfr[4][2];
#define TNR_NT_BARRIER(K) __builtin_amdgcn_s_barrier()
#define TNR_PP_SEG_END(K)                                           \
    __builtin_amdgcn_sched_barrier(0);                              \
    TNR_NT_BARRIER(K);                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
    __builtin_amdgcn_sched_barrier(0);                              \
    __builtin_amdgcn_s_setprio(1)
#define TNR_PP_MFMA_END(K)                                          \
    __builtin_amdgcn_s_setprio(0);                                  \
    __builtin_amdgcn_sched_barrier(0);                              \
    TNR_NT_BARRIER(K);                                              \
    __builtin_amdgcn_sched_barrier(0)
    constexpr bool reads_on = true, mfma_on = true;
  while (true) {
    int bm, bn, rs;
    bool tall_v;
    tile_coords(tile, nbm, nbn, g.gm, bm, bn);
    pp_panel(bm, nbm, g.mix_x, BM, rs, tall_v);
    const bool tall = __builtin_amdgcn_readfirstlane(tall_v ? 1 : 0) != 0;   // wave-uniform: scalar branches in the K loop
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K tile 0 (and 1's A0) of this tile were issued by the prologue -- before the previous tile's epilogue for all but
    // the first tile, so only that epilogue's own (younger) operations may still be in flight here
    // (a counted wait that leaves the previous epilogue's stores in flight -- vmcnt(number of stores the epilogue issued
    // after the prologue), full tiles only -- measured the same as waiting for everything both with the first epilogue and with
    // this one (tools/gemm_ab.py, +-0.3 % per shape), so the simple form stays)
    // A wait the compiler can see, unlike the K loop's, and on every path from the epilogue to K tile 0 (its state merge at
    // the loop head keeps an operation pending that any path leaves pending, so the wait has to sit here, not at the loop's
    // bottom): instances whose epilogue loads into registers - residual, auxiliary operand - otherwise get a vmcnt(0) of the
    // compiler's own in front of K tile 0's first MFMA, i.e. behind phase 0's LDS-DMA issues, once per tile.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();                        // K tile 0 is in LDS
    unsigned qn;                                         // lane 0 of wave 0: the next tile, in flight during this K loop
    pp_q_fetch(qn, g.queue + xcd * PP_Q_STRIDE, w == 0);
    if (wm == 1) __builtin_amdgcn_s_barrier();           // the stagger: group 1 runs one interval behind
    // One K tile, written once for both parities of the B register roles: bfr[BL], bfr[BL + 1] hold B columns 0-31 of this K
    // tile and bfr[BH], bfr[BH + 1] columns 32-63; the NEXT K tile's columns 0-31 are read in phase 3 into the BH pair (free
    // after quadrant M2), which is that tile's BL pair - so phase 0 reads A only (8 instead of 12 ds_read_b128 per wave) and the
    // per-phase counts are 8 / 4 / 8 / 4.  Worth -1.4 % over the 12 / 4 / 8 / 0 schedule (two-library A/B, tools/gemm_ab_lib.py;
    // 10 of 12 shapes faster), far less than the 9-10 % a probe with phase 0's B reads simply removed suggested: the total LDS
    // read volume, not its peak, is what the loop pays for.  Orders tried: B0 | B1, A1 | - | A0 (B1 lands too late for the
    // phase-2 wait on long K: +1.3 %), B two K tiles ahead from phase 3 (-0.6 %), A1 in phase 0 behind the B half tiles (six
    // LDS-DMA issues in the 8-read phase: +3.0 % against this order).
    // LDS-DMA per phase: B0(kt+1), B1(kt+1) | A1(kt+1) | - | A0(kt+2).  Waits: end of phase 2 vmcnt(2) = all but A1(kt+1)
    // -> B(kt+1) has landed one barrier (two for the staggered group) before phase 3 reads it; end of phase 3 vmcnt(2) =
    // all but A0(kt+2) -> A1(kt+1) landed before the next K tile.  WAR: B(kt-1) was last read in phase 1 of K tile kt-1 (its
    // columns 0-31 in phase 3 of kt-2), A1(kt-1) in group 1's phase 2 of kt-1, A0(kt) in group 0's phase 2 of kt.
#define TNR_PP_DUMMY_STORE(P)
#define TNR_PP_DUMMY_WAIT(P)                                                                                     \
    if ((P) == 2) { if (more && a_live) TNR_WAIT_VMCNT(2); else TNR_WAIT_VMCNT(0); }                             \
    else { if (kt + 2 < nk && a_live) TNR_WAIT_VMCNT(2); else TNR_WAIT_VMCNT(0); }
#define TNR_PP_KTILE(KT, BL, BH)                                                                                 \
    {                                                                                                            \
        const int kt = (KT);                                                                                     \
        const char* sa = smem + (kt & 1) * STAGE3 + wm * TILE_BYTES;                                             \
        const char* sb = smem + (kt & 1) * STAGE3 + (2 + (wn >> 1)) * TILE_BYTES + ((wn & 1) * 64) * 128;        \
        const char* sbn = smem + ((kt + 1) & 1) * STAGE3 + (2 + (wn >> 1)) * TILE_BYTES + ((wn & 1) * 64) * 128;  \
        const bool more = kt + 1 < nk;                                                                           \
        /* ---- phase 0: A rows 0-63 (+ B columns 0-31 of the first K tile) ; quadrant (lo, lo) */               \
        if (reads_on) _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                            \
            if (kt == 0)                                                                                         \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) bfr[BL + j][s] = *(const bf16x8*)(sb + j * 4 * 128 + boff[s]); \
            _Pragma("unroll") for (int i = 0; i < ILO; ++i) af[i][s] = *(const bf16x8*)(sa + i * 16 * 128 + foff[s]); \
        }                                                                                                        \
        if (more) { issue(2, kt + 1); issue(3, kt + 1); }                                                        \
        TNR_PP_DUMMY_STORE(0);                                                                                   \
        TNR_PP_SEG_END(0);                                                                                        \
        if (mfma_on) _Pragma("unroll") for (int s = 0; s < 2; ++s)                                               \
            _Pragma("unroll") for (int i = 0; i < ILO; ++i)                                                      \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
                    acc[i][j] = TNR_MFMA_16x16x32(bfr[BL + j][s], af[i][s], acc[i][j], 0, 0, 0);                 \
        TNR_PP_MFMA_END(1);                                                                                       \
        /* ---- phase 1: B columns 32-63 ; quadrant (lo, hi) */                                                  \
        if (reads_on) _Pragma("unroll") for (int s = 0; s < 2; ++s)                                              \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) bfr[BH + j][s] = *(const bf16x8*)(sb + (32 + j * 4) * 128 + boff[s]); \
        if (more) issue(1, kt + 1);                                                                              \
        TNR_PP_DUMMY_STORE(1);                                                                                   \
        TNR_PP_SEG_END(2);                                                                                        \
        if (mfma_on) _Pragma("unroll") for (int s = 0; s < 2; ++s)                                               \
            _Pragma("unroll") for (int i = 0; i < ILO; ++i)                                                      \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
                    acc[i][2 + j] = TNR_MFMA_16x16x32(bfr[BH + j][s], af[i][s], acc[i][2 + j], 0, 0, 0);         \
        TNR_PP_MFMA_END(3);                                                                                       \
        /* ---- phase 2: A rows 64-127 ; quadrant (hi, hi) ; B of K tile kt+1 must have landed before phase 3 reads it */ \
        if (reads_on) _Pragma("unroll") for (int s = 0; s < 2; ++s)                                              \
            _Pragma("unroll") for (int i = 0; i < IHI; ++i)                                                      \
                if (i + 1 < IHI || tall) af[i][s] = *(const bf16x8*)(sa + (ILO + i) * 16 * 128 + foff[s]);       \
        TNR_PP_DUMMY_WAIT(2);                                                                                    \
        TNR_PP_DUMMY_STORE(2);                                                                                   \
        TNR_PP_SEG_END(4);                                                                                        \
        if (mfma_on) _Pragma("unroll") for (int s = 0; s < 2; ++s)                                               \
            _Pragma("unroll") for (int i = 0; i < IHI; ++i)                                                      \
                if (i + 1 < IHI || tall) _Pragma("unroll") for (int j = 0; j < 2; ++j)                           \
                    acc[ILO + i][2 + j] = TNR_MFMA_16x16x32(bfr[BH + j][s], af[i][s], acc[ILO + i][2 + j], 0, 0, 0); \
        TNR_PP_MFMA_END(5);                                                                                       \
        /* ---- phase 3: B columns 0-31 of K tile kt+1 into the BH pair ; quadrant (hi, lo) ; A1(kt+1) must have landed */ \
        if (more) {                                                                                              \
            if (reads_on) _Pragma("unroll") for (int s = 0; s < 2; ++s)                                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) bfr[BH + j][s] = *(const bf16x8*)(sbn + j * 4 * 128 + boff[s]); \
            if (kt + 2 < nk) issue(0, kt + 2);                                                                   \
            TNR_PP_DUMMY_WAIT(3);                                                                                \
        }                                                                                                        \
        TNR_PP_DUMMY_STORE(3);                                                                                   \
        TNR_PP_SEG_END(6);                                                                                        \
        if (mfma_on) _Pragma("unroll") for (int s = 0; s < 2; ++s)                                               \
            _Pragma("unroll") for (int i = 0; i < IHI; ++i)                                                      \
                if (i + 1 < IHI || tall) _Pragma("unroll") for (int j = 0; j < 2; ++j)                           \
                    acc[ILO + i][j] = TNR_MFMA_16x16x32(bfr[BL + j][s], af[i][s], acc[ILO + i][j], 0, 0, 0);     \
        TNR_PP_MFMA_END(7);                                                                                       \
    }
    for (int kt2 = 0; kt2 < nk; kt2 += 2) {              // two K tiles per trip: the B register roles alternate at compile time
        TNR_PP_KTILE(kt2, 0, 2)
        if (kt2 + 1 < nk) TNR_PP_KTILE(kt2 + 1, 2, 0)
    }
#undef TNR_PP_KTILE
#undef TNR_PP_DUMMY_STORE
#undef TNR_PP_DUMMY_WAIT
#undef TNR_PP_SEG_END
#undef TNR_PP_MFMA_END
    pp_q_wait(qn);                                       // (the K loop's last waits were vmcnt(0) already)
    if (tid == 0) {
        qlds[1] = c0 + (int)qn;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();           // group 0 waits for group 1's last MFMA segment
    int next = qlds[1];                                  // behind that barrier (group 1: behind its last MFMA segment's) for everyone
    if (next >= c1) next = -1;
    // every fragment read of this tile has completed (group 1's last MFMA segment is behind the barrier above): the stage
    // ring is free, so the next tile's first loads go out BEFORE this tile's epilogue and land under it
    if (next >= 0) prologue(next, par ^ 1);
    nt_epilogue_cols<MI, CF>(g, acc, lut, bias_lds + par * 256, bm, bn, rs, tall ? MI : MI - 1, wm, wn, lane);
    par ^= 1;
    if (g.clock && next < 0 && tid == 0 && (int)blockIdx.x < g.clock_n) {   // shader cycles / 100 MHz ticks of this workgroup's life
        unsigned long long* o = g.clock + 2 * blockIdx.x;
        o[0] = __builtin_amdgcn_s_memtime() - pt0;
        o[1] = __builtin_amdgcn_s_memrealtime() - pr0;
    }
    if (next < 0) break;
    tile = next;
  }
  leave();
}

// wgrad v3: output tile 256 (n) x 256 (k); stage = [dY cols 0-127 | dY cols 128-255 | X cols 0-127 | X cols 128-255]
__global__ __launch_bounds__(512, 2) void gemm_tn256x256_kernel(TNArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wn = w >> 2, wk = w & 3;          // wave tile: 128 n x 64 k
    const int nbk = g.K >> 8;
    const int ntile = (g.N >> 8) * nbk;
    const int wg = xcd_remap(blockIdx.x, ntile * g.splits);
    const int z = wg / ntile, tile = wg - z * ntile;
    const int bn = tile / nbk, bk = tile - bn * nbk;
    const int mt0 = z * g.tiles_per_split;
    int mt1 = mt0 + g.tiles_per_split;
    if (mt1 > g.Mt) mt1 = g.Mt;
    const int nt = mt1 - mt0;

    const bf16* src[8];
    int64_t ldsrc[8];
    int dst[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int p = w * 8 + q, sub = p >> 4, pp = p & 15;
        int row = pp * 4 + (lane >> 4);
        int chunk = (lane & 15) ^ tn_swz(row);
        if (sub < 2) {
            src[q] = g.dY + (int64_t)(mt0 * 64 + row) * g.lddy + bn * 256 + sub * 128 + chunk * 8;
            ldsrc[q] = 64 * g.lddy;
        } else {
            src[q] = g.X + (int64_t)(mt0 * 64 + row) * g.ldx + bk * 256 + (sub - 2) * 128 + chunk * 8;
            ldsrc[q] = 64 * g.ldx;
        }
        dst[q] = sub * TILE_BYTES + pp * 1024;
    }
    auto stage = [&](int buf, int t) {
        char* base = smem + buf * STAGE3;
#pragma unroll
        for (int q = 0; q < 8; ++q) glds16(src[q] + (int64_t)t * ldsrc[q], base + dst[q]);
    };
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    int roff[2][2], rswz[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int row = 32 * s + 8 * g16 + q4 + 4 * h;
            roff[s][h] = row * 256 + (p4 & 1) * 8;
            rswz[s][h] = tn_swz(row);
        }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nt > 0) {
        stage(0, 0);
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            TNR_WAIT_VMCNT(0);
            __builtin_amdgcn_s_barrier();
            if (t + 1 < nt) stage(cur ^ 1, t + 1);
            const char* sy = smem + cur * STAGE3 + wn * TILE_BYTES;
            const char* sx = smem + cur * STAGE3 + (2 + (wk >> 1)) * TILE_BYTES;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 yf[8], xf[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    int cx = 2 * ((wk & 1) * 4 + tt) + (p4 >> 1);
                    bf16x4 x0 = ds_read_tr16(sx + roff[s][0] + ((cx ^ rswz[s][0]) << 4));
                    bf16x4 x1 = ds_read_tr16(sx + roff[s][1] + ((cx ^ rswz[s][1]) << 4));
                    xf[tt] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) {
                    int cy = 2 * tt + (p4 >> 1);
                    bf16x4 y0 = ds_read_tr16(sy + roff[s][0] + ((cy ^ rswz[s][0]) << 4));
                    bf16x4 y1 = ds_read_tr16(sy + roff[s][1] + ((cy ^ rswz[s][1]) << 4));
                    yf[tt] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = TNR_MFMA_16x16x32(xf[j], yf[i], acc[i][j], 0, 0, 0);
            }
        }
    }
    float* slab = g.ws + (int64_t)z * g.N * g.K;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int n = bn * 256 + wn * 128 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int k = bk * 256 + wk * 64 + j * 16 + (lane >> 4) * 4;
            *(f32x4*)(slab + (int64_t)n * g.K + k) = acc[i][j];
        }
    }
}


// wgrad v5 "register-staged": the v4 tile, (split, tile) units, queue and ping-pong of two wave groups, with a different path from
// memory to the MFMA operands and a different cut of the m step into phases.  Every MFMA gets the operand registers v4 gives it
// and every accumulator sees half 0 of an m step before half 1, so the slabs are bit-identical to v4's.
// What the round-5 barrier-arrival stamps showed (tools/tn_stamps.py, EXPERIMENTS item 32) and what follows from it:
//   * beside the other group's MFMAs a vector instruction of a load segment -- LDS read, vector-memory, plain VALU alike -- gets
//     one issue slot per MFMA, ~16 clk; in the wave's OWN MFMA segment it costs its 4-13 clk, nothing is free.  v4's load
//     segments carry 48 ds_read_b64_tr_b16 + 8 LDS-DMA issues (~100 clk each) + address arithmetic per wave and step;
//   * a CU accepts vector-memory instructions at ~30 B / clk: all the loads of a step issued by four waves in one segment (v4's
//     phase 3: 24 KB; the first register-staged form: 32 KB) hold that segment for 1 000-1 200 clk;
//   * every barrier-to-barrier interval costs ~100 clk on top of its MFMAs.
// Hence:
//   * operands through registers: a lane loads, one m step ahead, the 16-byte rows of an 8-column strip -- 4 rows of each 32-row
//     half of the step ("set A", "set B", 4 buffer_load_dwordx4 each) --, transposes a set with 16 v_perm_b32 into its 8 bytes of
//     8 fragments "one column, eight rows" of the 16x16x32 operand and stores them with 8 ds_write_b64 into a FRAGMENT-READY
//     image: fragment (operand, 16-column block, 32-row half) = 1 KB, lane-linear for the reading lane (g = lane >> 4: row
//     octet, r = lane & 15: column): 24 ds_read_b128 per wave and step instead of 48 transposing reads, no LDS-DMA, and every
//     LDS address a register + immediate.  16-byte slot of (g, r) in a fragment:  16 g + 8 (r >> 3) + ((r & 7) ^ key),
//     key = (block & 3) | ((r >> 3) << 2): the 16 lanes of a ds_write_b64 cycle (8 column strips = 8 keys, x the two 4-row
//     halves of a row octet) cover the 128-byte bank row once, the 16 lanes of a ds_read_b128 cycle the 256-byte one
//     (tools/scratch/tn_rs_image_check.py walks both maps);
//   * the transposition and the stores ride in the wave's own MFMA segments (one v_perm_b32 behind each of 16 MFMAs), the address
//     updates too; a load segment is fragment reads + 4 loads and nothing else;
//   * two phases per m step instead of four: phase S = 12 fragment reads of 32-row half S (X 4, dY 8) + the 4 loads of the set
//     the MFMA segment before has stored | 32 MFMAs (8 dY blocks x 4 X blocks) + the staging of set S of step t + 1.  A load has
//     two intervals (~0.9 us) to land; 16 KB of loads per interval.
// Hazards.  Stage (t + 1) & 1 receives half 0 of step t + 1 in the MFMA segments A of step t and half 1 in the segments B; group 1
// runs one interval behind group 0 (intervals of a step: group 0 L_A 0, M_A 1, L_B 2, M_B 3; group 1 L_A 1, M_A 2, L_B 3,
// M_B 4 = 0 of the next step); a load segment ends with lgkmcnt(0) BEFORE its barrier, so a wave's LDS operations of an interval
// are complete when the next but one begins at the latest (stores of an MFMA segment: by the end of the wave's next load segment).
//   RAW  half 0 of step t + 1: stored in intervals 1-2 of step t, complete by the end of interval 3, first read in interval 0 of
//        step t + 1.  half 1: stored in intervals 3-4, complete by the end of interval 1 of step t + 1, first read in its interval 2.
//   WAR  half 0 of step t - 1 (same stage): last read in interval 1 of step t - 1 ... overwritten from interval 1 of step t;
//        half 1: last read in interval 3 of step t - 1, overwritten from interval 3 of step t.
constexpr int RS_OP = 2 * TILE_BYTES;        // dY image | X image, 32 KB each per stage
constexpr int RS_LDS = RING3 + 64;
#define TNR_RS_BARRIER(K) __builtin_amdgcn_s_barrier()
__global__ __launch_bounds__(512, 2) void gemm_tn_rs_kernel(TNGroup grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w >> 2, wk = w & 3;          // wave tile: 128 n x 64 k
    unsigned* const queue = grp.queue;
    const int xcd = blockIdx.x & 7;
    int c0 = grp.xb[0], c1 = grp.xb[1];
#pragma unroll
    for (int x = 1; x < 8; ++x)
        if (xcd == x) { c0 = grp.xb[x]; c1 = grp.xb[x + 1]; }
    int* const qlds = (int*)(smem + RING3);
    unsigned q0;
    pp_q_fetch(q0, queue + xcd * PP_Q_STRIDE, w == 0);
    auto leave = [&]() {                                 // last workgroup out zeroes the counters for the next launch
        if (tid == 0) {
            const unsigned d = __hip_atomic_fetch_add(queue + 8 * PP_Q_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == gridDim.x - 1)
                for (int i = 0; i < 9; ++i)
                    __hip_atomic_store(queue + i * PP_Q_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // reader side: slot of this lane in a fragment, for the four values of block & 3
    int raddr[4];
    {
        const int r16 = lane & 15, g4 = lane >> 4, rhi = r16 >> 3, rl = (r16 & 7) ^ (rhi << 2);
#pragma unroll
        for (int v = 0; v < 4; ++v) raddr[v] = (g4 * 16 + rhi * 8 + (rl ^ v)) * 16;
    }
    // staging side: waves 0-3 carry dY, 4-7 X; wave q of the four owns m octet q of BOTH 32-row halves; a lane owns rows 4 sh ..
    // 4 sh + 3 of the octet (sh = lane bit 3) and column octet sc (of the tile's 32): 8 consecutive lanes load one 128-byte line
    // and the 16 lanes of a ds_write_b64 cycle cover 8 slots x 2 halves = the whole 128-byte bank row
    const int sop = w >> 2, sq = w & 3, sh = (lane >> 3) & 1, sc = ((lane >> 4) << 3) | (lane & 7);
    const int waddr = ((sop * RS_OP + (((sc >> 1) * 2) * 64 + sq * 16 + (sc & 1) * 8) * 16) ^
                       (((((sc >> 1) & 3) | ((sc & 1) << 2))) << 4)) + 8 * sh;     // half 0; half 1 = + 1024
    pp_q_wait(q0);
    if (tid == 0) qlds[0] = c0 + (int)q0;
    __syncthreads();
    int unit = qlds[0];
    if (unit >= c1) { leave(); return; }                 // whole workgroup, before any other barrier
  while (true) {
    const int su = __builtin_amdgcn_readfirstlane(unit);
    TNArgs g = grp.p[0];
    int lu = su;
    if (su >= grp.ubase[1]) { g = grp.p[1]; lu = su - grp.ubase[1]; }
    if (su >= grp.ubase[2]) { g = grp.p[2]; lu = su - grp.ubase[2]; }
    if (su >= grp.ubase[3]) { g = grp.p[3]; lu = su - grp.ubase[3]; }
    const int nbk = g.K >> 8;
    const int ntile = (g.N >> 8) * nbk;
    const int z = lu / ntile, tile = lu - z * ntile;
    const int bn = tile / nbk, bk = tile - bn * nbk;
    const int mt0 = z * g.tiles_per_split;
    int mt1 = mt0 + g.tiles_per_split;
    if (mt1 > g.Mt) mt1 = g.Mt;
    const int nk = mt1 - mt0;
    unsigned qn;                                         // lane 0 of wave 0: the next unit, in flight during this one (oldest operation)
    pp_q_fetch(qn, queue + xcd * PP_Q_STRIDE, w == 0);

    // this wave's operand as a raw buffer that starts at (first row of the unit, first column of the tile) and ends with the unit's
    // last row: the loads the pipeline issues for the two steps past the end return zeros without touching memory
    const int ld = sop ? (int)g.ldx : (int)g.lddy;
    const bf16* const opnd = sop ? g.X + ((int64_t)mt0 * 64 * g.ldx + bk * 256) : g.dY + ((int64_t)mt0 * 64 * g.lddy + bn * 256);
    const int64_t extent = ((int64_t)nk * 64 * ld - (sop ? bk : bn) * 256) * 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)opnd, 0, (int)(extent > 0x7fffffff ? 0x7fffffff : extent), 0x00020000);
    const int voff = ((sq * 8 + sh * 4) * ld + sc * 8) * 2;
    const int row_b = ld * 2, step_b = 64 * ld * 2;
    u32x4 in[8];                                // set A = the lane's 4 rows in half 0 of an m step, set B = in half 1
    auto fetch = [&](int t, int half) {         // 4 rows of half `half` of m step t -> registers
#pragma unroll
        for (int e = 0; e < 4; ++e)
            in[4 * half + e] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, t * step_b + (32 * half + e) * row_b, 0);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nk > 0) {
// end of a load segment: the wave's LDS operations complete BEFORE the barrier (the other group reads this wave's stores of the MFMA
// segment before, and overwrites what it has just read, right behind barriers; see Hazards)
#define TNR_PP_SEG_END(K)                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
    __builtin_amdgcn_sched_barrier(0);                              \
    TNR_RS_BARRIER(K);                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    __builtin_amdgcn_s_setprio(1)
#define TNR_PP_MFMA_END(K)                                          \
    __builtin_amdgcn_s_setprio(0);                                  \
    __builtin_amdgcn_sched_barrier(0);                              \
    TNR_RS_BARRIER(K);                                              \
    __builtin_amdgcn_sched_barrier(0)
// the segment's 32 MFMAs (8 dY blocks x 4 X blocks over 32-row half S); behind each of the first 16 one v_perm_b32 of the
// transposition of register set S (step t + 1's rows of half S -> the lane's 8 bytes of 8 fragments), after every fourth the two
// ds_write_b64 of a dword column; with S == 1 the 16 LDS addresses move to the other stage behind the last 16 MFMAs
#define TNR_RS_MFMAS(S)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                  \
        u32x2 lo, hi;                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
            acc[i][j] = TNR_MFMA_16x16x32(xf[j], yf[i], acc[i][j], 0, 0, 0);                                         \
            if (i < 4) {                                  /* MFMA row i carries dword column i of the block */       \
                const int e = 4 * (S) + 2 * (j & 1);                                                                 \
                if (j < 2) lo[j & 1] = __builtin_amdgcn_perm(in[e + 1][i], in[e][i], 0x05040100u);                   \
                else hi[j & 1] = __builtin_amdgcn_perm(in[e + 1][i], in[e][i], 0x07060302u);                         \
            } else if ((S) == 1 && i < 6) {                                                                          \
                if (j < 2) ray[2 * (i - 4) + j] += dlt; else rax[2 * (i - 4) + j - 2] += dlt;                        \
            } else if ((S) == 1) {                                                                                   \
                wa[4 * (i - 6) + j] -= dlt;                                                                          \
            }                                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        if (i < 4) {                                                                                                 \
            TNR_LDS(u32x2, wa[2 * i] + 1024 * (S)) = lo;                                                             \
            TNR_LDS(u32x2, wa[2 * i + 1] + 1024 * (S)) = hi;                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
    }
#define TNR_RS_READS(S)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) xf[j] = TNR_LDS(const bf16x8, rax[j] + (j * 2 + (S)) * 1024);      \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) yf[i] = TNR_LDS(const bf16x8, ray[i & 3] + (i * 2 + (S)) * 1024);  \
    __builtin_amdgcn_sched_barrier(0)
        bf16x8 yf[8], xf[4];
        unsigned ray[4], rax[4], wa[8];                     // LDS addresses: fragment reads (dY / X, by block & 3), the 8 stores
        const unsigned lds0 = lds_addr_of(smem);
        unsigned dlt = STAGE3;                              // reads move stage 0 -> 1 -> 0 ..., the stores of step t + 1 the other way
#pragma unroll
        for (int v = 0; v < 4; ++v) { ray[v] = lds0 + raddr[v] + wn * (8 * 2048); rax[v] = lds0 + raddr[v] + RS_OP + wk * (4 * 2048); }
#pragma unroll
        for (int f = 0; f < 8; ++f) wa[f] = lds0 + (waddr ^ (f << 4));
        {                                                   // step 0 -> stage 0 through registers the loop does not hold yet, so that
            u32x4 p0[8];                                    // set A of step 1 is in flight beside it (a unit of 30 steps notices)
#pragma unroll
            for (int e = 0; e < 8; ++e) p0[e] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (32 * (e >> 2) + (e & 3)) * row_b, 0);
            fetch(1, 0);                                    // (set B of step 1 is loaded in the first load segment)
#pragma unroll
            for (int S = 0; S < 2; ++S)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u32x2 lo, hi;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        lo[q] = __builtin_amdgcn_perm(p0[4 * S + 2 * q + 1][i], p0[4 * S + 2 * q][i], 0x05040100u);
                        hi[q] = __builtin_amdgcn_perm(p0[4 * S + 2 * q + 1][i], p0[4 * S + 2 * q][i], 0x07060302u);
                    }
                    TNR_LDS(u32x2, wa[2 * i] + 1024 * S) = lo;
                    TNR_LDS(u32x2, wa[2 * i + 1] + 1024 * S) = hi;
                }
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) wa[f] += STAGE3;        // the loop's stores of step t go to stage (t + 1) & 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // m step 0 is in LDS
        if (wn == 1) __builtin_amdgcn_s_barrier();          // the stagger: group 1 runs one interval behind
        for (int t = 0; t < nk; ++t) {
            TNR_RS_READS(0);
            fetch(t + 1, 1);                                // set B went to LDS in the MFMA segment before
            TNR_PP_SEG_END(0);
            TNR_RS_MFMAS(0);                                // + set A (half 0 of step t + 1) -> LDS
            TNR_PP_MFMA_END(1);
            TNR_RS_READS(1);
            fetch(t + 2, 0);
            TNR_PP_SEG_END(2);
            TNR_RS_MFMAS(1);                                // + set B (half 1 of step t + 1) -> LDS, addresses to the other stage
            dlt = 0u - dlt;
            TNR_PP_MFMA_END(3);
        }
        if (wn == 0) __builtin_amdgcn_s_barrier();          // all waves execute the same number of barriers
#undef TNR_RS_READS
#undef TNR_RS_MFMAS
#undef TNR_PP_SEG_END
#undef TNR_PP_MFMA_END
    }
    pp_q_wait(qn);
    if (tid == 0) qlds[1] = c0 + (int)qn;
    float* slab = g.ws + (int64_t)z * g.N * g.K;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int n = bn * 256 + wn * 128 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int k = bk * 256 + wk * 64 + j * 16 + (lane >> 4) * 4;
            *(f32x4*)(slab + (int64_t)n * g.K + k) = acc[i][j];
        }
    }
    __syncthreads();                                     // qlds[1] visible ; every fragment read of this unit is long complete
    const int next = qlds[1];
    if (next >= c1) break;
    unit = next;
    __syncthreads();                                     // qlds[1] is rewritten only after everybody has read it
  }
    leave();
}

__global__ void slab_reduce_kernel(const float* __restrict__ ws, int splits, int64_t NK, int K, float* out,
                                   int64_t ldo, int accumulate, float out_scale) {
    int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= NK) return;
    f32x4 s = *(const f32x4*)(ws + i4);
    for (int z = 1; z < splits; ++z) s += *(const f32x4*)(ws + (int64_t)z * NK + i4);
    int64_t n = i4 / K, k = i4 - n * K;
    float* o = out + n * ldo + k;
    s *= out_scale;
    if (accumulate) s += *(const f32x4*)o;
    *(f32x4*)o = s;
}

// the slab sums of up to four problems in one launch (blockIdx.y = problem): each element exactly as slab_reduce_kernel does it
struct SlabGroup {
    const float* ws[TN_MAXP]; float* out[TN_MAXP]; int64_t NK[TN_MAXP], ldo[TN_MAXP];
    int splits[TN_MAXP], K[TN_MAXP], accumulate[TN_MAXP]; float out_scale[TN_MAXP];
};
__global__ void slab_reduce_group_kernel(SlabGroup g) {
    const int pi = blockIdx.y;
    const float* ws = g.ws[0]; float* out = g.out[0]; int64_t NK = g.NK[0], ldo = g.ldo[0];
    int splits = g.splits[0], K = g.K[0], accumulate = g.accumulate[0]; float out_scale = g.out_scale[0];
#pragma unroll
    for (int k = 1; k < TN_MAXP; ++k)
        if (pi == k) {
            ws = g.ws[k]; out = g.out[k]; NK = g.NK[k]; ldo = g.ldo[k];
            splits = g.splits[k]; K = g.K[k]; accumulate = g.accumulate[k]; out_scale = g.out_scale[k];
        }
    int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= NK) return;
    f32x4 s = *(const f32x4*)(ws + i4);
    for (int z = 1; z < splits; ++z) s += *(const f32x4*)(ws + (int64_t)z * NK + i4);
    int64_t n = i4 / K, k = i4 - n * K;
    float* o = out + n * ldo + k;
    s *= out_scale;
    if (accumulate) s += *(const f32x4*)o;
    *(f32x4*)o = s;
}

}  // namespace

extern "C" int TNR_NAME(tnr_gemm_nt_do)(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                              int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres,
                              void* aux, int64_t ldaux, int flags, float* colsum_part, const tnr_dropout_t* drop,
                              void* stream);

extern "C" int TNR_NAME(tnr_gemm_nt)(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                           int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres,
                           void* aux, int64_t ldaux, int flags, void* stream) {
    return TNR_NAME(tnr_gemm_nt_ex)(A, lda, B, ldb, C, ldc, M, N, K, bias, res, ldres, aux, ldaux, flags, nullptr, stream);
}

extern "C" int64_t TNR_NAME(tnr_gemm_colsum_rows)(int64_t M) { return ((M + 255) / 256) * 4; }

// ---- routing ---------------------------------------------------------------------------------------
// Which kernel a launch takes depends on the shape only (and on the process-wide options of api.cpp, which tools set
// through tnr_gemm_set_option -- the library never reads the environment).  tnr_gemm_nt_route() exposes the decision
// so that the parity tests can pin every route.
static int device_cus() {
    if (tnr_gemm_opts()->cus > 0) return tnr_gemm_opts()->cus;
    static int cus[64] = {0};
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return 256;
    if (cus[devid] == 0) {
        hipDeviceProp_t prop;
        int n = 256;
        if (hipGetDeviceProperties(&prop, devid) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
        cus[devid] = n;
    }
    return cus[devid];
}

// Tiling of a ping-pong launch: instance (MI = 8: panels of 256 / 224 rows, MI = 7: 224 / 192), P row panels, x of them tall
// (pp_panel).  Cost model: a tile costs its rows + a fixed 24 (prologue latency, bias, queue), every XCD label's 1/8 of the tiles
// is pulled by 1/8 of the workgroups, a mixed launch pays half the height difference for the luck of the draw; candidates
// are all P between "all tall" and "all short".  mix = 0 (option `mix`, or column sums riding along: their partial rows
// are counted per 256-row panel, tnr_gemm_colsum_rows): the uniform tiling with the old 224 / 256 rule.
struct PpPlan { int mi, P, x; };
static PpPlan pp_plan(int64_t M, int64_t N, int flags, int n_cu) {
    const TnrGemmOpts& o = *tnr_gemm_opts();
    const int64_t ncol = N / 256;
    if ((flags & TNR_EPI_COLSUM) || !o.mix || !o.pp) {
        const int64_t t256 = ((M + 255) / 256) * ncol, t224 = ((M + 223) / 224) * ncol;
        const int64_t c256 = ((t256 + n_cu - 1) / n_cu) * 256, c224 = ((t224 + n_cu - 1) / n_cu) * 224;
        bool use224 = c224 * 108 < c256 * 100 && !(flags & TNR_EPI_COLSUM);   // per-tile fixed costs: need a clear win
        if (o.bm) use224 = o.bm == 224 && !(flags & TNR_EPI_COLSUM);
        const int P = (int)(use224 ? (M + 223) / 224 : (M + 255) / 256);
        return PpPlan{use224 ? 7 : 8, P, P};
    }
    PpPlan best{8, (int)((M + 255) / 256), (int)((M + 255) / 256)};
    double best_span = 1e30;
    const int64_t W = n_cu >= 8 ? n_cu / 8 : 1;
    for (int mi = 8; mi >= 7; --mi) {
        if (o.bm && o.bm != 32 * mi) continue;
        const int tall = 32 * mi, shrt = tall - 32;
        const int64_t pmin = (M + tall - 1) / tall, pmax = (M + shrt - 1) / shrt;
        for (int64_t p = pmin; p <= pmax; ++p) {
            int64_t x = M - p * shrt;
            x = x > 0 ? (x + 31) / 32 : 0;                          // tall panels needed to cover M rows
            const double ct = tall + 24.0, cs = shrt + 24.0, f = (double)x / (double)p, cbar = f * ct + (1.0 - f) * cs;
            const int64_t n = (p * ncol + 7) / 8, k = n / W, r = n % W;
            double span = (double)k * cbar + (r ? cbar : 0.0) + (x > 0 && x < p ? 0.5 * (ct - cs) : 0.0);
            if (k == 0) span = x > 0 ? ct : cs;
            if (span < best_span - 1e-9) { best_span = span; best = PpPlan{mi, (int)p, (int)x}; }
        }
    }
    return best;
}

static int nt_route(int64_t M, int64_t N, int64_t K, int flags, int n_cu) {
    const TnrGemmOpts& o = *tnr_gemm_opts();
    // short inputs (stage-1 title / body passes, small eval batches): when the 256x256 grid would leave more than 40 % of
    // the CUs without a tile, the 128x128 kernel (2 workgroups per CU) spreads the same work four times finer
    const bool sparse256 = (N % 256) == 0 && ((M + 255) / 256) * (N / 256) * 100 < (int64_t)n_cu * o.fine_pct && !(flags & TNR_EPI_COLSUM);
    const bool odd_gelu = (N % 256) != 0 && (flags & (TNR_EPI_GELU | TNR_EPI_MULDGELU));   // the 256x128 kernel has no table GELU
    if (o.ver == 1 || M <= 128 || odd_gelu || (sparse256 && o.allow_fine)) return TNR_ROUTE_128x128;
    if (o.ver == 2 || (N % 256) != 0) return TNR_ROUTE_256x128;
    return pp_plan(M, N, flags, n_cu).mi == 7 ? TNR_ROUTE_224x256 : TNR_ROUTE_256x256;
}

// the epilogue flag combinations of engine.py get an instance each with the flags at compile time (forward: QKV / pooled query,
// attention output + FFN down, FFN up with and without the pre-activation side output, pooling fc1 ; backward: the four dgrads)
#define TNR_PP_FLAG_SETS(X)                                                                                              \
    X(0) X(TNR_EPI_BIAS) X(TNR_EPI_RES) X(TNR_EPI_BIAS | TNR_EPI_RES) X(TNR_EPI_BIAS | TNR_EPI_GELU)                      \
    X(TNR_EPI_BIAS | TNR_EPI_GELU | TNR_EPI_AUXOUT) X(TNR_EPI_MULDGELU) X(TNR_EPI_MULDGELU | TNR_EPI_COLSUM)              \
    X(TNR_EPI_BIAS | TNR_EPI_TANH | TNR_EPI_OUTF32) X(TNR_EPI_BIAS | TNR_EPI_RES | TNR_EPI_DROPOUT)
// Counter sets of the ping-pong kernel's tile queue (the ONE piece of device state the library keeps, include/tnr_hip.h):
// 128 sets in a __device__ array, one per (device, stream) the kernel has been launched on -- launches of a stream run in order,
// so each finds the set its predecessor returned to zero, and launches of different streams never share one.  A set is zeroed
// by a hipMemsetAsync on its stream when the stream is first bound and again by tnr_gemm_queue_reset(); every launch leaves
// it at zero (the last workgroup out resets it).  The table never drains a device and never changes the current device: when
// it is full the launch is refused (TNR_EUNSUPPORTED) and the caller either reuses fewer streams or runs with option "pp" = 0.
// ONE table for both builds of this file (bf16 and -DTNR_BUILD_F16): it is defined in the bf16 translation unit and the fp16
// one calls into it (tnr_pp_queue_of, declared in common.h), so a stream that launches kernels of both builds is bound once and
// tnr_gemm_queue_reset reaches the counters whichever build's kernel was aborted.
constexpr int PP_LDS = LDS3_BYTES + 64;
[[maybe_unused]] constexpr int PP_QUEUE_SETS = 128;
#ifdef TNR_BUILD_F16
static unsigned* pp_queue_of(hipStream_t st, bool reset = false) { return tnr_pp_queue_of(st, reset); }
#else
__device__ unsigned g_pp_queue[PP_QUEUE_SETS * PP_Q_SET];
static unsigned* pp_queue_of(hipStream_t st, bool reset = false) { return tnr_pp_queue_of(st, reset); }
unsigned* tnr_pp_queue_of(void* stream, bool reset) {
    hipStream_t st = (hipStream_t)stream;
    struct Slot { int dev; hipStream_t st; };
    static std::mutex mu;
    static Slot slots[PP_QUEUE_SETS];
    static int nslot = 0;
    static unsigned* base[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { tnr_set_error("tnr_gemm_nt: no current device"); return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    if (!base[dev] && hipGetSymbolAddress((void**)&base[dev], HIP_SYMBOL(g_pp_queue)) != hipSuccess) {
        tnr_set_error("tnr_gemm_nt: tile-queue symbol not found");
        return nullptr;
    }
    unsigned* set = nullptr;
    for (int i = 0; i < nslot && !set; ++i)
        if (slots[i].dev == dev && slots[i].st == st) set = base[dev] + i * PP_Q_SET;
    const bool fresh = !set;
    if (!set) {
        if (nslot == PP_QUEUE_SETS) {
            tnr_set_error("tnr_gemm_nt: more than %d (device, stream) pairs have launched the persistent GEMM in this process "
                          "(reuse streams, or tnr_gemm_set_option(\"pp\", 0) for the kernel without a tile queue)", PP_QUEUE_SETS);
            return nullptr;
        }
        slots[nslot] = Slot{dev, st};
        set = base[dev] + (nslot++) * PP_Q_SET;
    }
    if ((fresh || reset) && hipMemsetAsync(set, 0, PP_Q_SET * sizeof(unsigned), st) != hipSuccess) {
        tnr_set_error("tnr_gemm_nt: could not zero the tile-queue counters");
        return nullptr;
    }
    return set;
}
#endif

#ifndef TNR_BUILD_F16
// Zero the calling stream's tile-queue counters (stream-ordered).  Only needed after a launch on that stream was aborted (device
// fault, process-level recovery): a completed launch always leaves them at zero.
extern "C" int tnr_gemm_queue_reset(void* stream) {
    return pp_queue_of((hipStream_t)stream, true) ? TNR_OK : TNR_EUNSUPPORTED;
}
#endif

template <int MI>
static void pp_launch(const NTArgs& g, unsigned grid, hipStream_t st) {
    switch (g.flags) {
#define TNR_PP_CASE(CF) \
    case (CF): hipLaunchKernelGGL((gemm_nt_pp_kernel<MI, (CF)>), dim3(grid), dim3(512), PP_LDS, st, g); break;
        TNR_PP_FLAG_SETS(TNR_PP_CASE)
#undef TNR_PP_CASE
    default: hipLaunchKernelGGL((gemm_nt_pp_kernel<MI, -1>), dim3(grid), dim3(512), PP_LDS, st, g); break;
    }
}

#ifndef TNR_BUILD_F16
// host-only (no HIP call): the tiling the persistent kernel would use on a device with n_cu compute units
extern "C" int tnr_gemm_nt_plan(int64_t M, int64_t N, int flags, int n_cu, int* mi, int* panels, int* tall) {
    TNR_CHECK_ARG(M >= 1 && N >= 256 && (N % 256) == 0 && n_cu >= 1 && mi && panels && tall, "tnr_gemm_nt_plan: bad argument");
    const PpPlan pl = pp_plan(M, N, flags, n_cu);
    *mi = pl.mi; *panels = pl.P; *tall = pl.x;
    return TNR_OK;
}
#endif

extern "C" int TNR_NAME(tnr_gemm_nt_route)(int64_t M, int64_t N, int64_t K, int flags) {
    return nt_route(M, N, K, flags, device_cus());
}

extern "C" int TNR_NAME(tnr_gemm_nt_ex)(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                              int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres,
                              void* aux, int64_t ldaux, int flags, float* colsum_part, void* stream) {
    return TNR_NAME(tnr_gemm_nt_do)(A, lda, B, ldb, C, ldc, M, N, K, bias, res, ldres, aux, ldaux, flags, colsum_part, nullptr, stream);
}

extern "C" int TNR_NAME(tnr_gemm_nt_do)(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                              int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres,
                              void* aux, int64_t ldaux, int flags, float* colsum_part, const tnr_dropout_t* drop,
                              void* stream) {
    TnrDrop dd;
    if (int rc = tnr_make_drop(drop, &dd, "tnr_gemm_nt")) return rc;
    TNR_CHECK_ARG(!(flags & TNR_EPI_DROPOUT), "tnr_gemm_nt: TNR_EPI_DROPOUT is set by passing a dropout site");
    if (dd.thresh) flags |= TNR_EPI_DROPOUT;
    TNR_CHECK_ARG(A && B && C, "tnr_gemm_nt: null operand");
    TNR_CHECK_ARG(M >= 1 && N >= 128 && (N % 128) == 0 && K >= 64 && (K % 64) == 0,
                  "tnr_gemm_nt: need N%%128==0, K%%64==0 (M=%ld N=%ld K=%ld)", (long)M, (long)N, (long)K);
    TNR_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldc % 4) == 0 && lda >= K && ldb >= K && ldc >= N,
                  "tnr_gemm_nt: bad leading dimension");
    TNR_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
                  "tnr_gemm_nt: operands must be 16-byte aligned");
    TNR_CHECK_ARG(!(flags & TNR_EPI_BIAS) || bias, "tnr_gemm_nt: TNR_EPI_BIAS without bias");
    TNR_CHECK_ARG(!(flags & TNR_EPI_RES) || (res && (ldres % 4) == 0), "tnr_gemm_nt: TNR_EPI_RES without res");
    TNR_CHECK_ARG(!(flags & (TNR_EPI_MULDGELU | TNR_EPI_AUXOUT)) || (aux && (ldaux % 4) == 0),
                  "tnr_gemm_nt: aux required");
    TNR_CHECK_ARG(M < (1 << 24), "tnr_gemm_nt: M too large");
    TNR_CHECK_ARG(!(flags & TNR_EPI_COLSUM) || (colsum_part && !(flags & TNR_EPI_OUTF32) && M > 128),
                  "tnr_gemm_nt: TNR_EPI_COLSUM needs a partial buffer, bf16 output and M > 128");
    const TnrGemmOpts& o = *tnr_gemm_opts();
    NTArgs g{(const bf16*)A, lda, (const bf16*)B, ldb, C, ldc, (int)M, (int)N, (int)K, bias,
             (const bf16*)res, ldres, (bf16*)aux, ldaux, flags, colsum_part, o.gm > 0 ? o.gm : 8, o.nt, 0, dd, 0, 0, nullptr, 0, nullptr, 0};
    g.clock = (unsigned long long*)o.clock_buf;
    g.clock_n = o.clock_n;
#define TNR_PP_ATTR(CF)                                                                                                     \
        (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<8, CF>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS); \
        (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<7, CF>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
    TNR_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES + LUT_N * 8);
        (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING2);
        (void)hipFuncSetAttribute((const void*)gemm_nt256x256_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt256x256_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3_BYTES);
        TNR_PP_FLAG_SETS(TNR_PP_ATTR)
        TNR_PP_ATTR(-1)
    });
#undef TNR_PP_ATTR
    hipStream_t st = (hipStream_t)stream;
    const int n_cu = device_cus();
    switch (nt_route(M, N, K, flags, n_cu)) {
    case TNR_ROUTE_128x128:
        hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)(((M + 127) / 128) * (N / 128))), dim3(256), 2 * BUF_BYTES + LUT_N * 8, st, g);
        break;
    case TNR_ROUTE_256x128:
        hipLaunchKernelGGL(gemm_nt256_kernel, dim3((unsigned)(((M + 255) / 256) * (N / 128))), dim3(512), RING2, st, g);
        break;
    case TNR_ROUTE_224x256:
        if (o.pp) {
            const PpPlan pl = pp_plan(M, N, flags, n_cu);
            g.mix_p = pl.P; g.mix_x = pl.x;
            if (!(g.queue = pp_queue_of(st))) return TNR_EUNSUPPORTED;
            pp_launch<7>(g, (unsigned)std::min<int64_t>((int64_t)pl.P * (N / 256), std::max(n_cu, 8)), st);   // >= 8: every XCD label needs a workgroup
        }
        else hipLaunchKernelGGL((gemm_nt256x256_kernel<7>), dim3((unsigned)(((M + 223) / 224) * (N / 256))), dim3(512), LDS3_BYTES, st, g);
        break;
    default:
        if (o.pp) {
            const PpPlan pl = pp_plan(M, N, flags, n_cu);
            g.mix_p = pl.P; g.mix_x = pl.x;
            if (!(g.queue = pp_queue_of(st))) return TNR_EUNSUPPORTED;
            pp_launch<8>(g, (unsigned)std::min<int64_t>((int64_t)pl.P * (N / 256), std::max(n_cu, 8)), st);
        }
        else hipLaunchKernelGGL((gemm_nt256x256_kernel<8>), dim3((unsigned)(((M + 255) / 256) * (N / 256))), dim3(512), LDS3_BYTES, st, g);
        break;
    }
    TNR_CHECK_LAUNCH("tnr_gemm_nt");
    return TNR_OK;
}

// (tiles_per_split + 2) m steps of 64 rows x the larger leading dimension x 2 bytes must fit a 31-bit offset (gemm_tn_rs_kernel)
static bool tn_rs_unit_fits(int tps, int64_t lddy, int64_t ldx) {
    return ((int64_t)tps + 2) * 64 * std::max(lddy, ldx) * 2 < ((int64_t)1 << 31);
}

extern "C" int64_t TNR_NAME(tnr_gemm_tn_ws_elems)(int64_t N, int64_t K, int splits) { return N * K * (int64_t)splits; }

extern "C" int TNR_NAME(tnr_gemm_tn_wgrad_ex)(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW,
                                    int64_t lddw, int64_t M, int64_t N, int64_t K, float* ws, int splits,
                                    int accumulate, float out_scale, void* stream) {
    TNR_CHECK_ARG(dY && X && dW && ws, "tnr_gemm_tn_wgrad: null operand");
    TNR_CHECK_ARG(M >= 1 && (N % 128) == 0 && (K % 128) == 0 && N >= 128 && K >= 128,
                  "tnr_gemm_tn_wgrad: need N%%128==0, K%%128==0 (N=%ld K=%ld)", (long)N, (long)K);
    TNR_CHECK_ARG((lddy % 8) == 0 && (ldx % 8) == 0 && (lddw % 4) == 0 && lddy >= N && ldx >= K && lddw >= K,
                  "tnr_gemm_tn_wgrad: bad leading dimension");
    TNR_CHECK_ARG(((uintptr_t)dY % 16) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)dW % 16) == 0 &&
                      ((uintptr_t)ws % 16) == 0, "tnr_gemm_tn_wgrad: operands must be 16-byte aligned");
    int Mt = (int)((M + 63) / 64);
    TNR_CHECK_ARG(splits >= 1 && splits <= 64, "tnr_gemm_tn_wgrad: splits out of range");
    if (splits > Mt) splits = Mt;
    int tps = (Mt + splits - 1) / splits;
    splits = (Mt + tps - 1) / tps;
    TNArgs g{(const bf16*)dY, lddy, (const bf16*)X, ldx, ws, Mt, (int)N, (int)K, tps, splits, nullptr};
    const int ver = tnr_gemm_opts()->ver;
    // gemm_tn_rs_kernel addresses a unit's rows through a raw buffer with 32-bit byte offsets (its pipeline also issues the loads
    // of two m steps past the unit's end and relies on them falling OUTSIDE the buffer): a unit must stay under 2 GiB per operand
    TNR_CHECK_ARG(tn_rs_unit_fits(tps, lddy, ldx),
                  "tnr_gemm_tn_wgrad: %d rows per split x leading dimension %ld exceed the 2 GiB a unit may span - raise `splits`",
                  tps * 64, (long)std::max(lddy, ldx));
    TNR_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING2);
        (void)hipFuncSetAttribute((const void*)gemm_tn256x256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING3);
        (void)hipFuncSetAttribute((const void*)gemm_tn_rs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
    });
    if (ver == 1 || (N % 256) != 0) {
        dim3 grid((unsigned)((N / 128) * (K / 128) * splits));
        hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 2 * BUF_BYTES, (hipStream_t)stream, g);
    } else if (ver == 2 || (K % 256) != 0) {
        dim3 grid((unsigned)((N / 256) * (K / 128) * splits));
        hipLaunchKernelGGL(gemm_tn256_kernel, grid, dim3(512), RING2, (hipStream_t)stream, g);
    } else {
        dim3 grid((unsigned)((N / 256) * (K / 256) * splits));
        if (tnr_gemm_opts()->tnpp) {
            // persistent: at most one workgroup per CU (and at least one per XCD label) pulls the (split, tile) units
            TNGroup grp{};
            grp.p[0] = g;
            for (int i = 1; i <= TN_MAXP; ++i) grp.ubase[i] = (int)grid.x;
            tn_group_ranges(grp, 1);
            if (!(grp.queue = pp_queue_of((hipStream_t)stream))) return TNR_EUNSUPPORTED;
            const int n_cu = device_cus();
            dim3 pgrid((unsigned)std::min<int64_t>((int64_t)grid.x, std::max(n_cu, 8)));
            hipLaunchKernelGGL(gemm_tn_rs_kernel, pgrid, dim3(512), RS_LDS, (hipStream_t)stream, grp);
        }
        else hipLaunchKernelGGL(gemm_tn256x256_kernel, grid, dim3(512), RING3, (hipStream_t)stream, g);
    }
    TNR_CHECK_LAUNCH("tnr_gemm_tn_wgrad");
    int64_t NK = N * K;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((NK / 4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)ws, splits, NK, (int)K, dW, lddw, accumulate, out_scale);
    TNR_CHECK_LAUNCH("tnr_gemm_tn_wgrad/reduce");
    return TNR_OK;
}

extern "C" int TNR_NAME(tnr_gemm_tn_wgrad_group)(const tnr_wgrad_problem_t* p, int n, void* stream) {
    TNR_CHECK_ARG(p && n >= 1 && n <= TN_MAXP, "tnr_gemm_tn_wgrad_group: 1 .. %d problems", TN_MAXP);
    // accumulate == 2: this problem CONTINUES the sum of its predecessor - another row range (other operands, other M) of the same
    // gradient: same dW, N, K, lddw, out_scale; its slabs follow the predecessor's in the HEAD problem's workspace and one slab
    // sum covers the chain (stage 1: the title pass and the body pass of one Linear)
    for (int i = 0; i < n; ++i)
        if (p[i].accumulate == 2)
            TNR_CHECK_ARG(i > 0 && p[i].dW == p[i - 1].dW && p[i].N == p[i - 1].N && p[i].K == p[i - 1].K && p[i].lddw == p[i - 1].lddw &&
                              p[i].out_scale == p[i - 1].out_scale,
                          "tnr_gemm_tn_wgrad_group: problem %d continues problem %d but differs from it in dW / N / K / lddw / out_scale", i, i - 1);
    auto head_of = [&](int i) { while (p[i].accumulate == 2) --i; return i; };
    bool pp = tnr_gemm_opts()->tnpp && tnr_gemm_opts()->ver == 3 && n > 1;
    for (int i = 0; i < n; ++i) pp = pp && p[i].N >= 256 && p[i].K >= 256 && (p[i].N % 256) == 0 && (p[i].K % 256) == 0;
    if (!pp) {                                 // a shape off the persistent kernel's route (or one problem): one launch each, the same results
        for (int i = 0; i < n; ++i) {
            // (a chained problem - accumulate 2 - simply adds to what its predecessor wrote; the head's workspace serves both in turn)
            int rc = TNR_NAME(tnr_gemm_tn_wgrad_ex)(p[i].dY, p[i].lddy, p[i].X, p[i].ldx, p[i].dW, p[i].lddw, p[i].M, p[i].N, p[i].K,
                                                     p[i].accumulate == 2 ? p[head_of(i)].ws : p[i].ws, p[i].splits,
                                                     p[i].accumulate == 2 ? 1 : p[i].accumulate, p[i].out_scale, stream);
            if (rc != TNR_OK) return rc;
        }
        return TNR_OK;
    }
    TNGroup grp{};
    SlabGroup sg{};
    int64_t units = 0, maxblk = 0;
    for (int i = 0; i < n; ++i) {
        const tnr_wgrad_problem_t& q = p[i];
        const bool chained = q.accumulate == 2;
        TNR_CHECK_ARG(q.dY && q.X && q.dW && (q.ws || chained), "tnr_gemm_tn_wgrad_group: null operand");
        TNR_CHECK_ARG(q.M >= 1 && (q.lddy % 8) == 0 && (q.ldx % 8) == 0 && (q.lddw % 4) == 0 && q.lddy >= q.N && q.ldx >= q.K && q.lddw >= q.K,
                      "tnr_gemm_tn_wgrad_group: bad shape / leading dimension (problem %d)", i);
        TNR_CHECK_ARG(((uintptr_t)q.dY % 16) == 0 && ((uintptr_t)q.X % 16) == 0 && ((uintptr_t)q.dW % 16) == 0 &&
                          (chained || ((uintptr_t)q.ws % 16) == 0),
                      "tnr_gemm_tn_wgrad_group: operands must be 16-byte aligned");
        TNR_CHECK_ARG(q.splits >= 1 && q.splits <= 64, "tnr_gemm_tn_wgrad_group: splits out of range");
        int Mt = (int)((q.M + 63) / 64), splits = q.splits;                       // the split arithmetic of tnr_gemm_tn_wgrad_ex
        if (splits > Mt) splits = Mt;
        const int tps = (Mt + splits - 1) / splits;
        splits = (Mt + tps - 1) / tps;
        TNR_CHECK_ARG(tn_rs_unit_fits(tps, q.lddy, q.ldx),
                      "tnr_gemm_tn_wgrad_group: problem %d: %d rows per split x leading dimension %ld exceed the 2 GiB a unit may span - raise "
                      "`splits`", i, tps * 64, (long)std::max(q.lddy, q.ldx));
        const int h = head_of(i);
        float* const ws_i = chained ? (float*)sg.ws[h] + (int64_t)sg.splits[h] * q.N * q.K : q.ws;   // behind the chain's slabs so far
        grp.p[i] = TNArgs{(const bf16*)q.dY, q.lddy, (const bf16*)q.X, q.ldx, ws_i, Mt, (int)q.N, (int)q.K, tps, splits, nullptr};
        grp.ubase[i] = (int)units;
        units += (q.N / 256) * (q.K / 256) * splits;
        sg.ws[i] = ws_i; sg.out[i] = q.dW; sg.ldo[i] = q.lddw; sg.K[i] = (int)q.K; sg.out_scale[i] = q.out_scale;
        if (chained) {                      // its slabs are summed with the head's: no slab sum of its own
            sg.splits[h] += splits; sg.NK[i] = 0; sg.splits[i] = 0; sg.accumulate[i] = 0;
        } else {
            sg.NK[i] = q.N * q.K; sg.splits[i] = splits; sg.accumulate[i] = q.accumulate;
        }
        maxblk = std::max<int64_t>(maxblk, (q.N * q.K / 4 + 255) / 256);
    }
    if (units < 8) {
        // fewer units than XCD labels: the grid min(units, ...) would leave labels that tn_group_ranges gives work without a
        // workgroup (their units never computed, the slab sum reading unwritten slabs) -> one launch per problem, the same results
        for (int i = 0; i < n; ++i) {
            // (a chained problem - accumulate 2 - simply adds to what its predecessor wrote; the head's workspace serves both in turn)
            int rc = TNR_NAME(tnr_gemm_tn_wgrad_ex)(p[i].dY, p[i].lddy, p[i].X, p[i].ldx, p[i].dW, p[i].lddw, p[i].M, p[i].N, p[i].K,
                                                     p[i].accumulate == 2 ? p[head_of(i)].ws : p[i].ws, p[i].splits,
                                                     p[i].accumulate == 2 ? 1 : p[i].accumulate, p[i].out_scale, stream);
            if (rc != TNR_OK) return rc;
        }
        return TNR_OK;
    }
    for (int i = n; i <= TN_MAXP; ++i) grp.ubase[i] = (int)units;
    tn_group_ranges(grp, n);
    TNR_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute((const void*)gemm_tn_rs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
    });
    if (!(grp.queue = pp_queue_of((hipStream_t)stream))) return TNR_EUNSUPPORTED;
    const int n_cu = device_cus();
    const dim3 ggrid((unsigned)std::min<int64_t>(units, std::max(n_cu, 8)));
    hipLaunchKernelGGL(gemm_tn_rs_kernel, ggrid, dim3(512), RS_LDS, (hipStream_t)stream, grp);
    TNR_CHECK_LAUNCH("tnr_gemm_tn_wgrad_group");
    hipLaunchKernelGGL(slab_reduce_group_kernel, dim3((unsigned)maxblk, (unsigned)n), dim3(256), 0, (hipStream_t)stream, sg);
    TNR_CHECK_LAUNCH("tnr_gemm_tn_wgrad_group/reduce");
    return TNR_OK;
}


extern "C" int TNR_NAME(tnr_gemm_tn_wgrad)(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW,
                                 int64_t lddw, int64_t M, int64_t N, int64_t K, float* ws, int splits,
                                 int accumulate, void* stream) {
    return TNR_NAME(tnr_gemm_tn_wgrad_ex)(dY, lddy, X, ldx, dW, lddw, M, N, K, ws, splits, accumulate, 1.0f, stream);
}
