// Probe: gfx950 LDS-DMA with 12 bytes per lane (buffer_load_dwordx3 ... lds) -- wanted for the gathered 24-byte value rows of
// a 6-support layer (two lanes per row).  Question: does lane l land at base + 12 l (lane-linear, 768 bytes per instruction),
// with gathered 4-byte-aligned sources and EXEC masking like the 16-byte form (probe_glds.hip)?
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_glds3.hip -o tools/probes/probe_glds3.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma12(u32x4 rs, uint32_t lds_addr, int voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx3 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}
__global__ __launch_bounds__(64) void k(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ out, int mode, int n) {
    __shared__ __attribute__((aligned(16))) float lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = -1.f;
    __syncthreads();
    const uint64_t a = reinterpret_cast<uint64_t>(src);
    const u32x4 rs = u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)n * 4u, 0x00020000u};
    const uint32_t l0 = (uint32_t)(uintptr_t)((lds_void*)lds);
    if (mode == 0) dma12(rs, l0, lane * 12);                                       // linear
    if (mode == 1) { if (lane & 1) dma12(rs, l0, lane * 12); }                     // exec-masked
    if (mode == 2) dma12(rs, l0 + 16, idx[lane >> 1] * 24 + 12 * (lane & 1) + 4);  // gathered 24-byte rows, 4-byte aligned, two lanes per row
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    const int n = 8192;
    std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
    std::vector<int> hi(32); for (int i = 0; i < 32; ++i) hi[i] = (i * 37 + 11) % 300;
    float *d, *o; int* di;
    hipMalloc(&d, n * 4); hipMalloc(&o, 512 * 4); hipMalloc(&di, 32 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(di, hi.data(), 32 * 4, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, di, o, mode, n);
        std::vector<float> r(512); hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
        std::vector<float> e(512, -1.f);
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 3; ++j) {
            if (mode == 0) e[l * 3 + j] = (float)(l * 3 + j);
            if (mode == 1 && (l & 1)) e[l * 3 + j] = (float)(l * 3 + j);
            if (mode == 2) e[4 + l * 3 + j] = (float)(hi[l >> 1] * 6 + 3 * (l & 1) + 1 + j);
        }
        int bad = 0;
        for (int i = 0; i < 512; ++i) if (r[i] != e[i]) { if (bad < 6) printf("mode %d word %d got %g want %g\n", mode, i, r[i], e[i]); ++bad; }
        printf("x3 mode %d: %s (%d mismatches)\n", mode, bad ? "FAIL" : "ok", bad);
        bad_total += bad;
    }
    return bad_total != 0;
}
