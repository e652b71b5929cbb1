// Probe (round 4): when does a queued v_mfma_f32_16x16x32_bf16 read its operands?  A wave issues a chain of N dependent MFMAs
// (acc += A.B, A = B = 1.0 -> +32 per MFMA) and IMMEDIATELY overwrites the B operand -- by an LDS load of 2.0s (asynchronous
// return) or by v_mov (VALU) -- then waits.  If every MFMA read B when it was ISSUED the result is 32 N; every MFMA that fetched
// its operand after the overwrite adds 64 instead.  hipcc assumes "at issue" and re-uses operand registers for the next loads at
// once.  Varied: chain length, waves per SIMD (other waves keep the matrix pipe busy), overwrite kind.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_mfma_war.hip -o tools/probes/probe_mfma_war.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N, int KIND>   // KIND 0: ds_read into B, 2: ds_read into A
__global__ void k(float* out, int busy) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lds[i] = 0x40004000u;          // bf16 2.0 pairs
    __syncthreads();
    u32x4 a = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};                // bf16 1.0 x 8
    u32x4 b = a;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned laddr = (unsigned)(lane * 16);
    if (wave == 0) {
        if (KIND == 0) {
            asm volatile(
                ".rept %c4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t.endr\n\t"
                "ds_read_b128 %2, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                : "+v"(acc), "+v"(a), "+v"(b) : "v"(laddr), "n"(N) : "memory");
        } else if (KIND == 2) {
            asm volatile(
                ".rept %c4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t.endr\n\t"
                "ds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                : "+v"(acc), "+v"(a), "+v"(b) : "v"(laddr), "n"(N) : "memory");
        }
        out[lane] = acc[0];
    } else if (busy) {
        // the other waves keep the matrix pipes of their SIMDs busy
        f32x4 c2 = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < 200; ++i)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
        if (c2[0] == 12345.f) out[64 + threadIdx.x] = c2[0];
    }
}

template <int N, int KIND>
void run(float* d, int threads, int busy, const char* what) {
    hipMemset(d, 0, 4096);
    hipLaunchKernelGGL((k<N, KIND>), dim3(1), dim3(threads), 0, 0, d, busy);
    float h[64];
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    float mn = h[0], mx = h[0];
    for (int i = 1; i < 64; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
    printf("%-22s N=%2d threads=%4d busy=%d: acc in [%g, %g], 32 N = %d%s\n", what, N, threads, busy, mn, mx, 32 * N,
           (mn == 32.f * N && mx == 32.f * N) ? "" : "   <-- operand fetched AFTER the overwrite");
}

int main() {
    float* d; hipMalloc(&d, 4096);
    // waves of one workgroup go to SIMDs round robin: wave 0 and wave 4 share SIMD 0 (320+ threads)
    for (int busy = 0; busy < 2; ++busy)
        for (int threads : {64, 320, 576}) {
            run<1, 0>(d, threads, busy, "ds_read into B"); run<2, 0>(d, threads, busy, "ds_read into B"); run<4, 0>(d, threads, busy, "ds_read into B");
            run<6, 0>(d, threads, busy, "ds_read into B"); run<12, 0>(d, threads, busy, "ds_read into B"); run<16, 0>(d, threads, busy, "ds_read into B");
            run<6, 2>(d, threads, busy, "ds_read into A"); run<12, 2>(d, threads, busy, "ds_read into A");
        }
    return 0;
}
