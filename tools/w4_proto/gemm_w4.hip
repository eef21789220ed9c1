// PROTOTYPE (tools only, never linked into libtnr_hip.so): the 256 x 256 x 64 NT GEMM tile with FOUR waves of 128 x 128
// (one per SIMD, 256 accumulator registers) and a software-pipelined K loop - the layout whose LDS traffic per K tile is
// 128 KB of fragment reads instead of the eight-wave kernel's 192 KB (EXPERIMENTS.md section 4, items 5 / 6).  Round 1's
// version of this layout ("v7", tools/retired/) read all 16 fragments of a k half and then issued its 64 MFMAs, leaving
// the interleave to the compiler: 20-40 % slower than eight waves.  Here the fragment reads of k half s+1 are placed
// between the MFMA rows of half s at source level and pinned with sched_barrier.  fp16 only, plain store, M % 256 == 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DTNR_BUILD_F16 -o libw4.so gemm_w4.hip
#include "../../tiny-newsrec_amd/csrc/common.h"

namespace {
constexpr int TILE_BYTES = 128 * 128;          // one operand half tile: 128 rows x 64 k x 2 B
constexpr int STAGE = 4 * TILE_BYTES;          // A0 | A1 | B0 | B1

struct W4Args { const bf16* A; int64_t lda; const bf16* B; int64_t ldb; bf16* C; int64_t ldc; int M, N, K; int variant; };

template <int VAR>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4_kernel(W4Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MI = 8, NJ = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int nbn = g.N >> 8, nbm = g.M >> 8;
    // same tile order as the product kernel: XCD-contiguous runs, groups of 8 row tiles x all column tiles
    const int nwg = nbm * nbn;
    int wg;
    {
        const int xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    int bm, bn;
    {
        const int GM = 8, per = GM * nbn, grp = wg / per, first = grp * GM;
        const int gsz = nbm - first < GM ? nbm - first : GM, local = wg - grp * per;
        bm = first + local % gsz;
        bn = local / gsz;
    }
    // staging: wave w issues the 16 one-KiB pieces (8 rows each) of half tile w
    const bf16* src0;
    int64_t row_stride8;
    {
        const int row = lane >> 3, chunk = (lane & 7) ^ (row & 7);
        if (w < 2) { src0 = g.A + (int64_t)(bm * 256 + w * 128 + row) * g.lda + chunk * 8; row_stride8 = 8 * g.lda; }
        else       { src0 = g.B + (int64_t)(bn * 256 + (w - 2) * 128 + row) * g.ldb + chunk * 8; row_stride8 = 8 * g.ldb; }
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE + w * TILE_BYTES;
#pragma unroll
        for (int q = 0; q < 16; ++q) glds16(src0 + q * row_stride8 + kt * 64, base + q * 1024);
    };
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = (lane & 15) * 128 + ((((4 * s) + (lane >> 4)) ^ (lane & 7)) << 4);

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = g.K >> 6;
    stage(0, 0);
    bf16x8 af[2][MI], bfr[2][NJ];
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (VAR == 2 && kt > 0) {                  // probe: MFMA issue only (fragments of K tile 0 reused)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = TNR_MFMA_16x16x32(bfr[s][j], af[s][i], acc[i][j], 0, 0, 0);
            continue;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk && VAR != 2 && VAR != 3) stage(cur ^ 1, kt + 1);
        const bool more = kt + 1 < nk;
        char* nbase = smem + (cur ^ 1) * STAGE + w * TILE_BYTES;
        auto piece = [&](int q) { glds16(src0 + q * row_stride8 + (kt + 1) * 64, nbase + q * 1024); };
        const char* sa = smem + cur * STAGE + wm * TILE_BYTES;
        const char* sb = smem + cur * STAGE + (2 + wn) * TILE_BYTES;
        // k half 0 fragments (exposed: the only reads of a K tile nothing runs under)
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[0][j] = *(const bf16x8*)(sb + j * 16 * 128 + foff[0]);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][i] = *(const bf16x8*)(sa + i * 16 * 128 + foff[0]);
        if (VAR == 3) {
            // as VAR 0, plus the next K tile's LDS-DMA issued one piece per MFMA row instead of 16 in a burst at the top
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                bfr[1][i] = *(const bf16x8*)(sb + i * 16 * 128 + foff[1]);
                af[1][i] = *(const bf16x8*)(sa + i * 16 * 128 + foff[1]);
                if (more) piece(i);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = TNR_MFMA_16x16x32(bfr[0][j], af[0][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (more) piece(8 + i);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = TNR_MFMA_16x16x32(bfr[1][j], af[1][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (VAR == 0 || VAR == 2) {
            // half 0 MFMAs with the reads of half 1 between the rows
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                bfr[1][i] = *(const bf16x8*)(sb + i * 16 * 128 + foff[1]);
                af[1][i] = *(const bf16x8*)(sa + i * 16 * 128 + foff[1]);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = TNR_MFMA_16x16x32(bfr[0][j], af[0][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = TNR_MFMA_16x16x32(bfr[1][j], af[1][i], acc[i][j], 0, 0, 0);
            }
        } else {
            // round 1's form: all reads of a half, then its MFMAs (the compiler schedules)
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[1][j] = *(const bf16x8*)(sb + j * 16 * 128 + foff[1]);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[1][i] = *(const bf16x8*)(sa + i * 16 * 128 + foff[1]);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = TNR_MFMA_16x16x32(bfr[s][j], af[s][i], acc[i][j], 0, 0, 0);
        }
    }
    // plain store: lane (m = lane & 15, q = lane >> 4) holds row 16 i + m, columns 16 j + 4 q .. + 3 of the wave's 128 x 128
    const int m16 = lane & 15, qd = lane >> 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        bf16* crow = g.C + (int64_t)(bm * 256 + wm * 128 + i * 16 + m16) * g.ldc + bn * 256 + wn * 128 + 4 * qd;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[i][j][r];
            *(bf16x4*)(crow + j * 16) = o;
        }
    }
}
}  // namespace

extern "C" int w4_gemm(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                       int variant, void* stream) {
    if ((M % 256) || (N % 256) || (K % 64)) return -1;
    W4Args g{(const bf16*)A, lda, (const bf16*)B, ldb, (bf16*)C, ldc, (int)M, (int)N, (int)K, variant};
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        (void)hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        (void)hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        (void)hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        attr = true;
    }
    const unsigned grid = (unsigned)((M / 256) * (N / 256));
    if (variant == 0) hipLaunchKernelGGL(gemm_nt_w4_kernel<0>, dim3(grid), dim3(256), 2 * STAGE, (hipStream_t)stream, g);
    else if (variant == 1) hipLaunchKernelGGL(gemm_nt_w4_kernel<1>, dim3(grid), dim3(256), 2 * STAGE, (hipStream_t)stream, g);
    else if (variant == 2) hipLaunchKernelGGL(gemm_nt_w4_kernel<2>, dim3(grid), dim3(256), 2 * STAGE, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(gemm_nt_w4_kernel<3>, dim3(grid), dim3(256), 2 * STAGE, (hipStream_t)stream, g);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
