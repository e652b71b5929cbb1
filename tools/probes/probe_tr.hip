// probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 values = their own element index; every lane passes
// the byte address addr[lane] (host-chosen) and gets 4 u16 back.  Prints lane -> the 4 element indices it received.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const int* addr, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x];
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
    out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
    int h[64]; uint16_t o[256];
    int* d; uint16_t* dout;
    hipMalloc(&d, 256); hipMalloc(&dout, 512);
    for (int mode = 0; mode < 3; ++mode) {
        for (int l = 0; l < 64; ++l) {
            const int t = l & 15, g = l >> 4;
            if (mode == 0) h[l] = l * 8;                                   // contiguous 8-byte chunks
            if (mode == 1) h[l] = ((t >> 2) * 64 + (t & 3) * 8) + g * 1024; // rows of 64 B: lane t -> row t>>2, chunk t&3
            if (mode == 2) h[l] = (t * 64) + g * 8;                        // lane t -> row t (stride 64 B), chunk g
        }
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
        hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %5d (elem %4d): %4d %4d %4d %4d\n", l, h[l], h[l] / 2, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
    }
    return 0;
}
