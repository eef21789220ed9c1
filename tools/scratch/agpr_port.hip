// Does an MFMA's result write-back share a register-file port with the LDS read returns of the OTHER wave on its SIMD, and does it
// matter whether the accumulators are VGPRs or AGPRs?  One 512-thread workgroup per CU: waves 0-3 issue back-to-back
// v_mfma_f32_16x16x32_f16 on 16 accumulators, waves 4-7 (their SIMD partners) stream ds_read_b128 (or nothing).  Prints cycles per
// MFMA (s_memtime) for {VGPR, AGPR accumulators} x {partner idle, partner reading LDS}.
//   hipcc -O3 --offload-arch=gfx950 tools/scratch/agpr_port.hip -o tools/scratch/agpr_port && tools/scratch/agpr_port
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;

template <bool AGPR, bool READS>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, int iters) {
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
    } else if (READS) {
        const char* p = lds + lane * 16;
        f4 sink = (f4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {          // 8 reads per 16 MFMAs: the v8 NT kernel's ratio beside a 16-MFMA segment
                f4 v = *(const f4*)(p + ((i * 1024 + it * 64) & 65535 & ~15));
                sink += v;
            }
        }
        if (sink[0] == 12345.678f) out[1] = 0;
    }
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 256 * 8 * 8);
    const int iters = 2000;
    std::vector<unsigned long long> h(256 * 8);
    auto run = [&](auto kern, const char* name) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, iters);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 4; ++w) { s += (double)h[b * 8 + w]; ++n; }
        printf("%-44s %.2f cycles per MFMA\n", name, s / n / (iters * 16.0));
    };
    run(k<false, false>, "accumulators in VGPRs, partner idle");
    run(k<false, true>, "accumulators in VGPRs, partner reads LDS");
    run(k<true, false>, "accumulators in AGPRs, partner idle");
    run(k<true, true>, "accumulators in AGPRs, partner reads LDS");
    return 0;
}
