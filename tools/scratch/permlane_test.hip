// prints the lane movement of v_permlane32_swap / v_permlane16_swap on gfx950 (tools only)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned lane = threadIdx.x;
    unsigned a = 1000 + lane, b = 2000 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + lane] = s[0]; out[192 + lane] = s[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"p32 r0(vdst=a)", "p32 r1(src=b)", "p16 r0(vdst=a)", "p16 r1(src=b)"};
    for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[t * 64 + l]); printf("\n"); }
    return 0;
}
