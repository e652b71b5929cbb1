#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned l = threadIdx.x;
    unsigned a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
    for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[t * 64 + l]); printf("\n"); }
    return 0;
}
