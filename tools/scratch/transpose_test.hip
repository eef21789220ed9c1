// checks the 4x4 lane/register transpose used by nt_epilogue_direct (tools only)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ void lane_swap32(float& a, float& b) {
    auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(t[0]); b = __uint_as_float(t[1]);
}
__device__ __forceinline__ void lane_swap16(float& a, float& b) {
    auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(t[0]); b = __uint_as_float(t[1]);
}
template <int MI>
__device__ __forceinline__ void epi(f32x4 (&acc)[MI][4], float* out, int lane) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        float R[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 t = acc[i][j];
            R[j][0] = t[0]; R[j][1] = t[1]; R[j][2] = t[2]; R[j][3] = t[3];
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            lane_swap32(R[0][d], R[2][d]);
            lane_swap32(R[1][d], R[3][d]);
            lane_swap16(R[0][d], R[1][d]);
            lane_swap16(R[2][d], R[3][d]);
        }
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = R[e >> 2][e & 3];
        // lane (m, q') now holds columns 16 q' .. 16 q' + 15 of row i*16 + m
        float* p = out + (i * 16 + (lane & 15)) * 64 + (lane >> 4) * 16;
#pragma unroll
        for (int e = 0; e < 16; e += 4) *(f32x4*)(p + e) = (f32x4){v[e], v[e + 1], v[e + 2], v[e + 3]};
    }
}
__global__ void k(float* out) {
    int lane = threadIdx.x;
    f32x4 acc[2][4];
    // accumulator layout: lane (m = lane & 15, q = lane >> 4), block j, element r  <->  row i*16 + m, col 16 j + 4 q + r
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (float)((i * 16 + (lane & 15)) * 100 + 16 * j + 4 * (lane >> 4) + r);
    epi<2>(acc, out, lane);
}
int main() {
    float* d; hipMalloc(&d, 32 * 64 * 4);
    k<<<1, 64>>>(d);
    static float h[32 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 64; ++c) if (h[r * 64 + c] != (float)(r * 100 + c)) { if (bad < 8) printf("bad [%d][%d] = %g\n", r, c, h[r * 64 + c]); ++bad; }
    printf("transpose: %d mismatches\n", bad);
    return 0;
}
