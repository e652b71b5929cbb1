// Probe of the gfx950 LDS-DMA (global_load_lds_dwordx4) semantics the fwd3 / bwd4 landing rings rely on:
//   (1) lane l of a wave-instruction lands at  base + 16 l  (lane-linear, 1 KiB per instruction);
//   (2) lanes switched off by EXEC write nothing and the others keep their lane positions;
//   (3) the per-lane SOURCE address may be any 4-byte aligned address (gathers, unaligned starts);
//   (4) a counted vmcnt + s_barrier makes other waves' DMA data readable.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_glds.hip -o /tmp/probe_glds ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)l, 16, 0, 0);
}
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ out, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 256 + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * 256 + 64; i += 256) lds[i] = -1.f;
    __syncthreads();
    float* dst = lds + wave * 256;
    if (mode == 0) dma16(src + wave * 256 + lane * 4, dst);                       // linear
    if (mode == 1) { if (lane < 33 && (lane & 1)) dma16(src + wave * 256 + lane * 4, dst); }   // exec-masked
    if (mode == 2) dma16(src + 1 + wave * 256 + lane * 4, dst);                    // 4-byte aligned source
    if (mode == 3) dma16(src + (size_t)idx[wave * 64 + lane] * 4 + 1, dst);        // gather, odd offsets
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int w2 = (wave + 1) & 3;                                                // read ANOTHER wave's landing zone
    for (int j = 0; j < 4; ++j) out[(w2 * 64 + lane) * 4 + j] = lds[w2 * 256 + lane * 4 + j];
}
int main() {
    const int n = 4096;
    std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
    std::vector<int> hi(256); for (int i = 0; i < 256; ++i) hi[i] = (i * 37 + 11) % 900;
    float *d, *o; int* di;
    hipMalloc(&d, n * 4); hipMalloc(&o, 1024 * 4); hipMalloc(&di, 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(di, hi.data(), 256 * 4, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, di, o, mode);
        std::vector<float> r(1024); hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            float e;
            if (mode == 0) e = (float)(w * 256 + l * 4 + j);
            else if (mode == 1) e = (l < 33 && (l & 1)) ? (float)(w * 256 + l * 4 + j) : -1.f;
            else if (mode == 2) e = (float)(1 + w * 256 + l * 4 + j);
            else e = (float)(hi[w * 64 + l] * 4 + 1 + j);
            if (r[(w * 64 + l) * 4 + j] != e) { if (bad < 4) printf("mode %d w %d l %d j %d got %g want %g\n", mode, w, l, j, r[(w * 64 + l) * 4 + j], e); ++bad; }
        }
        printf("mode %d: %s (%d mismatches)\n", mode, bad ? "FAIL" : "ok", bad);
        bad_total += bad;
    }
    return bad_total != 0;
}
