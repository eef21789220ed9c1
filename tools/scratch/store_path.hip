// Microbenchmark of a CU's store path (EXPERIMENTS.md section 4 item 15): 256 workgroups (one per CU) x 8 waves, every wave issues
// NST 16-byte-per-lane stores of registers it already holds, in one of several address patterns, into a buffer that stays in L2 /
// into a large one; reports shader cycles per store instruction and per CU.   hipcc --offload-arch=gfx950 -O3 store_path.hip -o store_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int NST = 16;
// pattern 0: 16 rows x 64 B per instruction (the ping-pong epilogue: lane (m = lane & 15, q = lane >> 4) -> row m, bytes 16 q .. +15)
// pattern 1: 2 rows x 512 B per instruction (the LDS-staged epilogue of round 1)
// pattern 2: 1 KB contiguous per instruction
// pattern 3: 64 rows x 16 B per instruction (row per lane)
// pattern 4: 8 rows x 128 B per instruction (full cache lines: what a lane-pair exchange in the epilogue would give)
// pattern 5: 4 rows x 256 B per instruction
// patterns 6-8: 8 rows x 128 B per instruction as the ping-pong epilogue can form them with a lane-pair exchange: lane (m = lane & 15,
//   q = lane >> 4) writes 16 bytes at column piece q (+ 4 for the pair's second lane) of the pair's lower row; pairs (m, m ^ 1) / (m, m ^ 4) / (m, m ^ 8)
template <int PAT>
__global__ __launch_bounds__(512) void k(char* out, long ld, long wg_stride, int reps, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    char* base = out + (long)blockIdx.x * wg_stride + (long)w * 128;        // wave w owns byte columns 128 w .. of each row
    u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            long off;
            if (PAT == 0) off = (long)(i * 16 + (lane & 15)) * ld + (lane >> 4) * 16;
            else if (PAT == 1) off = (long)(i * 2 + (lane >> 5)) * ld + (lane & 31) * 16 - (long)w * 128 + (long)w * 512 * 0 + 0;
            else if (PAT == 2) off = (long)i * ld + lane * 16 - (long)w * 128 + (long)w * ld * NST;
            else if (PAT == 6) off = (long)(i * 16 + (lane & 14)) * ld + (lane >> 4) * 16 + (lane & 1) * 64;
            else if (PAT == 7) off = (long)(i * 16 + (lane & 11)) * ld + (lane >> 4) * 16 + ((lane >> 2) & 1) * 64;
            else if (PAT == 8) off = (long)(i * 16 + (lane & 7)) * ld + (lane >> 4) * 16 + ((lane >> 3) & 1) * 64;
            else if (PAT == 4) off = (long)(i * 8 + (lane >> 3)) * ld + (lane & 7) * 16;
            else if (PAT == 5) off = (long)(i * 4 + (lane >> 4)) * ld + (lane & 15) * 16 - (long)w * 128 + (long)(w & 3) * 256 + (long)(w >> 2) * NST * 4 * ld;
            else off = (long)(i * 64 + lane) * ld;
            if (PAT == 1) off += (long)w * NST * 2 * ld;      // each wave its own rows, full 512-byte segments
            *(u32x4*)(base + off) = v;
            v[1] += 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}
int main() {
    const long ld = 1536;                       // bytes per row (768 16-bit columns)
    const long rows_per_wg = 64 * NST + 64, wg_stride = rows_per_wg * ld;
    char* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * wg_stride); hipMalloc(&cyc, 256 * 8);
    hipMemset(out, 0, 256 * wg_stride);
    const char* names[9] = {"16 rows x 64 B (ping-pong epilogue)", "2 rows x 512 B (LDS-staged epilogue)", "1 KB contiguous", "64 rows x 16 B (row per lane)",
                            "8 rows x 128 B (full lines)", "4 rows x 256 B", "pairs (m, m^1): 8 rows x 128 B", "pairs (m, m^4): 8 rows x 128 B", "pairs (m, m^8): 8 rows x 128 B"};
    for (int pat = 0; pat < 9; ++pat)
        for (int reps : {1, 8}) {
            std::vector<double> res;
            for (int it = 0; it < 5; ++it) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, ld, wg_stride, reps, cyc);
                if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, ld, wg_stride, reps, cyc);
                if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, ld, wg_stride, reps, cyc);
                if (pat == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, ld, wg_stride, reps, cyc);
                if (pat == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, ld, wg_stride, reps, cyc);
                if (pat == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, out, ld, wg_stride, reps, cyc);
                if (pat == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, out, ld, wg_stride, reps, cyc);
                if (pat == 7) hipLaunchKernelGGL(k<7>, dim3(256), dim3(512), 0, 0, out, ld, wg_stride, reps, cyc);
                if (pat == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 0, 0, out, ld, wg_stride, reps, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                std::vector<unsigned long long> h(256);
                hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
                std::sort(h.begin(), h.end());
                res.push_back((double)h[128]);
            }
            std::sort(res.begin(), res.end());
            const double c = res[2];
            printf("%-40s reps %d: %8.0f cycles per workgroup = %6.1f cycles per store instruction (128 x %d per CU), %5.1f B / clk / CU\n",
                   names[pat], reps, c, c / (8.0 * NST * reps), reps, 8.0 * NST * reps * 1024 / c);
        }
    return 0;
}
