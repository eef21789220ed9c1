// What slows an MFMA stream when its SIMD partner works?  One 512-thread workgroup per CU: waves 0-3 issue v_mfma_f32_16x16x32_f16
// (16 accumulators, back to back, priority 1); waves 4-7 - their SIMD partners - do one of: nothing, stream ds_read_b128, stream
// global_load_dwordx4 from an L2-resident buffer, stream LDS-DMA (global_load_lds_dwordx4), or a mix like a GEMM load segment
// (8 reads + 2 DMA per 16 partner MFMAs, barrier-free).  Variants of the MFMA wave: accumulators in VGPRs / AGPRs; operands constant
// or re-read from LDS before every group of 16 (ds_read_b128 x 2 + s_waitcnt).  Prints cycles per MFMA (s_memtime).
//   hipcc -O3 --offload-arch=gfx950 tools/scratch/agpr_port.hip -o tools/scratch/agpr_port && tools/scratch/agpr_port
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;

template <bool AGPR, int PARTNER, bool REREAD>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, const f4* gbuf, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 65536 / 4; i += 512) ((float*)lds)[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    if (w < 4) {
        h8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
        f4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            if (REREAD) {
                a = *(const h8*)(lds + lane * 16 + ((it * 2048) & 32767));
                b = *(const h8*)(lds + 32768 + lane * 16 + ((it * 2048) & 16383));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
        if (s == 12345.678f) out[0] = 0;
    } else if (PARTNER) {
        const char* p = lds + lane * 16;
        const f4* gp = gbuf + (blockIdx.x & 63) * 1024 + lane;      // 64 KB windows: L2-resident
        f4 sink = (f4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            if (PARTNER == 1 || PARTNER == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) sink += *(const f4*)(p + ((i * 1024 + it * 64) & 65535 & ~15));
            }
            if (PARTNER == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sink += __builtin_nontemporal_load(gp + ((i * 64 + it * 256) & 1023 & ~63));
            }
            if (PARTNER == 3 || PARTNER == 4) {
#pragma unroll
                for (int i = 0; i < (PARTNER == 3 ? 4 : 2); ++i)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gp + ((i * 64 + it * 256) & 1023 & ~63)),
                                                     (void __attribute__((address_space(3)))*)(lds + 49152 + (w - 4) * 4096 + i * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
        }
        if (PARTNER == 5 || PARTNER == 6) {                             // a VALU stream: 4 independent chains of v_add / v_perm
            unsigned v0 = lane, v1 = lane * 3, v2 = lane * 5, v3 = lane * 7;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < (PARTNER == 5 ? 16 : 4); ++i)
                    asm volatile("v_add_u32 %0, %0, %1\n v_perm_b32 %1, %1, %2, %3\n v_add_u32 %2, %2, %3\n v_perm_b32 %3, %3, %0, %1"
                                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
                if (PARTNER == 6) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) sink += *(const f4*)(p + ((i * 1024 + it * 64) & 65535 & ~15));
                }
            }
            if (v0 + v1 + v2 + v3 == 0x12345678u) out[2] = 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (sink[0] == 12345.678f) out[1] = 0;
    }
}

// 32 accumulators over 8 A x 4 B fragments in distinct registers (the GEMM kernels' shape); ORDER 0: A outer, B inner; 1: B outer
template <int ORDER, int PARTNER>
__global__ __launch_bounds__(512, 2) void k32(unsigned long long* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 65536 / 4; i += 512) ((float*)lds)[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    if (w < 4) {
        h8 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3;
        const _Float16 s = (_Float16)(0.01f * lane);
        a0 = (h8)(s); a1 = (h8)(s + (_Float16)1); a2 = (h8)(s + (_Float16)2); a3 = (h8)(s + (_Float16)3);
        a4 = (h8)(s + (_Float16)4); a5 = (h8)(s + (_Float16)5); a6 = (h8)(s + (_Float16)6); a7 = (h8)(s + (_Float16)7);
        b0 = (h8)(s * (_Float16)2); b1 = (h8)(s * (_Float16)3); b2 = (h8)(s * (_Float16)4); b3 = (h8)(s * (_Float16)5);
        f4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define MF(I, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[I]) : "v"(A), "v"(B))
        for (int it = 0; it < iters; ++it) {
            if (ORDER == 0) {
                MF(0, a0, b0); MF(1, a0, b1); MF(2, a0, b2); MF(3, a0, b3); MF(4, a1, b0); MF(5, a1, b1); MF(6, a1, b2); MF(7, a1, b3);
                MF(8, a2, b0); MF(9, a2, b1); MF(10, a2, b2); MF(11, a2, b3); MF(12, a3, b0); MF(13, a3, b1); MF(14, a3, b2); MF(15, a3, b3);
                MF(16, a4, b0); MF(17, a4, b1); MF(18, a4, b2); MF(19, a4, b3); MF(20, a5, b0); MF(21, a5, b1); MF(22, a5, b2); MF(23, a5, b3);
                MF(24, a6, b0); MF(25, a6, b1); MF(26, a6, b2); MF(27, a6, b3); MF(28, a7, b0); MF(29, a7, b1); MF(30, a7, b2); MF(31, a7, b3);
            } else {
                MF(0, a0, b0); MF(4, a1, b0); MF(8, a2, b0); MF(12, a3, b0); MF(16, a4, b0); MF(20, a5, b0); MF(24, a6, b0); MF(28, a7, b0);
                MF(1, a0, b1); MF(5, a1, b1); MF(9, a2, b1); MF(13, a3, b1); MF(17, a4, b1); MF(21, a5, b1); MF(25, a6, b1); MF(29, a7, b1);
                MF(2, a0, b2); MF(6, a1, b2); MF(10, a2, b2); MF(14, a3, b2); MF(18, a4, b2); MF(22, a5, b2); MF(26, a6, b2); MF(30, a7, b2);
                MF(3, a0, b3); MF(7, a1, b3); MF(11, a2, b3); MF(15, a3, b3); MF(19, a4, b3); MF(23, a5, b3); MF(27, a6, b3); MF(31, a7, b3);
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) r += acc[i][0] + acc[i][3];
        if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
        if (r == 12345.678f) out[0] = 0;
    } else if (PARTNER) {
        const char* p = lds + lane * 16;
        f4 sink = (f4){0.f, 0.f, 0.f, 0.f};
        unsigned v0 = lane, v1 = lane * 3, v2 = lane * 5, v3 = lane * 7;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < (PARTNER == 1 ? 0 : PARTNER == 5 ? 16 : 4); ++i)
                asm volatile("v_add_u32 %0, %0, %1\n v_perm_b32 %1, %1, %2, %3\n v_add_u32 %2, %2, %3\n v_perm_b32 %3, %3, %0, %1"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (PARTNER != 5) {
#pragma unroll
                for (int i = 0; i < 12; ++i) sink += *(const f4*)(p + ((i * 1024 + it * 64) & 65535 & ~15));
            }
        }
        if (v0 + v1 + v2 + v3 == 0x12345678u) out[2] = 0;
        if (sink[0] == 12345.678f) out[1] = 0;
    }
}

int main() {
    unsigned long long* d;
    f4* g;
    (void)hipMalloc(&d, 256 * 8 * 8);
    (void)hipMalloc(&g, 64 * 1024 * 16);
    (void)hipMemset(g, 0, 64 * 1024 * 16);
    const int iters = 2000;
    std::vector<unsigned long long> h(256 * 8);
    auto run = [&](auto kern, const char* name) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, g, iters);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, g, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 4; ++w) { s += (double)h[b * 8 + w]; ++n; }
        printf("%-72s %.2f cycles per MFMA\n", name, s / n / (iters * 16.0));
    };
    run(k<false, 0, false>, "VGPR acc, constant operands, partner idle");
    run(k<false, 1, false>, "VGPR acc, constant operands, partner: 8 ds_read_b128 per 16 MFMAs' time");
    run(k<false, 2, false>, "VGPR acc, constant operands, partner: global_load_dwordx4 stream");
    run(k<false, 3, false>, "VGPR acc, constant operands, partner: LDS-DMA stream");
    run(k<false, 4, false>, "VGPR acc, constant operands, partner: 8 ds_read + 2 LDS-DMA");
    run(k<false, 0, true>, "VGPR acc, operands re-read from LDS per 16 MFMAs, partner idle");
    run(k<false, 1, true>, "VGPR acc, operands re-read from LDS, partner: ds_read stream");
    run(k<false, 4, true>, "VGPR acc, operands re-read from LDS, partner: 8 ds_read + 2 LDS-DMA");
    run(k<false, 5, false>, "VGPR acc, constant operands, partner: VALU stream (64 per 16 MFMAs' loop)");
    run(k<false, 6, false>, "VGPR acc, constant operands, partner: 16 VALU + 8 ds_read_b128");
    auto run32 = [&](auto kern, const char* name) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, iters / 2);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, iters / 2);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 4; ++w) { s += (double)h[b * 8 + w]; ++n; }
        printf("%-72s %.2f cycles per MFMA\n", name, s / n / (iters / 2 * 32.0));
    };
    run32(k32<0, 0>, "32 acc, 8 A x 4 B fragments, A outer, partner idle");
    run32(k32<1, 0>, "32 acc, 8 A x 4 B fragments, B outer, partner idle");
    run32(k32<0, 1>, "32 acc, 8 A x 4 B fragments, A outer, partner: 12 ds_read_b128");
    run32(k32<0, 5>, "32 acc, 8 A x 4 B fragments, A outer, partner: VALU stream");
    run32(k32<0, 6>, "32 acc, 8 A x 4 B fragments, A outer, partner: 16 VALU + 12 ds_read");
    run(k<true, 0, false>, "AGPR acc, constant operands, partner idle");
    run(k<true, 4, true>, "AGPR acc, operands re-read from LDS, partner: 8 ds_read + 2 LDS-DMA");
    return 0;
}
