// Do VALU and MFMA instructions overlap on a gfx950 SIMD (same wave, other waves)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_overlap.hip -o /tmp/probe_overlap && /tmp/probe_overlap
// Inline asm, so that the instruction stream is exactly: NM x v_mfma_f32_16x16x32_bf16 on 4 independent AGPR accumulators
// and NV x v_fma_f32 on 8 independent VGPRs, either interleaved (1 MFMA : NV/NM VALU) or in two blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MF(ACC) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(a), "v"(b))
#define VF(X) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(X) : "v"(c))
#define VF4 VF(v0); VF(v1); VF(v2); VF(v3)
#define VF8 VF4; VF(v4); VF(v5); VF(v6); VF(v7)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7, c = 0.999f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f); b[i] = (__bf16)(i * 0.5f); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { MF(a0); MF(a1); MF(a2); MF(a3); MF(a0); MF(a1); MF(a2); MF(a3); }                     // 8 MFMA
        if (MODE == 1) { VF8; VF8; VF8; VF8; }                                                                  // 32 VALU
        if (MODE == 2) { MF(a0); MF(a1); MF(a2); MF(a3); MF(a0); MF(a1); MF(a2); MF(a3); VF8; VF8; VF8; VF8; }  // blocks
        if (MODE == 3) { MF(a0); VF4; MF(a1); VF4; MF(a2); VF4; MF(a3); VF4; MF(a0); VF4; MF(a1); VF4; MF(a2); VF4; MF(a3); VF4; }
        if (MODE == 4) { MF(a0); VF8; MF(a1); VF8; MF(a2); VF8; MF(a3); VF8; MF(a0); VF8; MF(a1); VF8; MF(a2); VF8; MF(a3); VF8; }  // 8 : 64
        if (MODE == 5) { MF(a0); VF(v0); VF(v1); MF(a1); VF(v2); VF(v3); MF(a2); VF(v4); VF(v5); MF(a3); VF(v6); VF(v7);
                         MF(a0); VF(v0); VF(v1); MF(a1); VF(v2); VF(v3); MF(a2); VF(v4); VF(v5); MF(a3); VF(v6); VF(v7); }         // 8 : 16
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int MODE>
void run(float* out, int wps, const char* what) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256 * wps, 256>>>(out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<256 * wps, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD=%d: %7.1f ns per SIMD-iteration, %6.1f ns per wave-iteration\n", what, wps, ms * 1e6 / iters, ms * 1e6 / iters / wps);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int wps : {1, 2, 4}) {
        run<0>(out, wps, "8 MFMA");
        run<1>(out, wps, "32 VALU");
        run<2>(out, wps, "8 MFMA then 32 VALU");
        run<3>(out, wps, "8 x (1 MFMA, 4 VALU)");
        run<4>(out, wps, "8 x (1 MFMA, 8 VALU)");
        run<5>(out, wps, "8 x (1 MFMA, 2 VALU)");
    }
    return 0;
}
