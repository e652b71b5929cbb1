// Probe (round 4), fixed registers: how long after its issue does v_mfma_f32_16x16x32_bf16 (and v_mfma_f32_16x16x16_bf16) still
// read its B operand?  acc = sum of N MFMAs with A = B = 1.0 (32 or 16 per MFMA).  Right after the last MFMA, D wait states later,
// ONE dword of B is overwritten with zeros -- dword `W` of the operand (0 = k slots 0-1 ... 3 = k slots 6-7 of every lane) -- by a
// VALU v_mov or by an LDS load.  A deficit in acc = that dword was fetched after the overwrite.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_mfma_war2.hip -o tools/probes/probe_mfma_war2.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K32, int W, int D, int LDS, int N = 1, int BUSY = 0>
__global__ void k(float* out) {
    extern __shared__ unsigned int lds[];                     // 4 KB of zeros, written below through asm-visible stores
    const int lane = threadIdx.x & 63;
    float res = 0.f;
    if (threadIdx.x < 64) {
        asm volatile(
            "v_mov_b32 v19, 0\n\tv_lshlrev_b32 v18, 2, %1\n\t"
            "ds_write_b32 v18, v19\n\tds_write_b32 v18, v19 offset:256\n\tds_write_b32 v18, v19 offset:512\n\tds_write_b32 v18, v19 offset:768\n\ts_waitcnt lgkmcnt(0)\n\t"
            "v_mov_b32 v20, 0x3f803f80\n\tv_mov_b32 v21, 0x3f803f80\n\tv_mov_b32 v22, 0x3f803f80\n\tv_mov_b32 v23, 0x3f803f80\n\t"
            "v_mov_b32 v24, 0x3f803f80\n\tv_mov_b32 v25, 0x3f803f80\n\tv_mov_b32 v26, 0x3f803f80\n\tv_mov_b32 v27, 0x3f803f80\n\t"
            "v_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"
            "s_nop 7\n\t"
            ".rept %c6\n\t"
            ".if %c2\n\tv_mfma_f32_16x16x32_bf16 v[28:31], v[20:23], v[24:27], v[28:31]\n\t.else\n\tv_mfma_f32_16x16x16_bf16 v[28:31], v[20:21], v[24:25], v[28:31]\n\t.endif\n\t"
            ".endr\n\t"
            ".if %c4 > 0\n\ts_nop %c4 - 1\n\t.endif\n\t"
            ".if %c5 == 2\n\tv_lshlrev_b32 v18, 2, v18\n\tds_read_b128 v[24:27], v18\n\ts_waitcnt lgkmcnt(0)\n\t.endif\n\t"
            ".if %c5 == 3\n\tds_read_b32 v28, v18\n\ts_waitcnt lgkmcnt(0)\n\t.endif\n\t"      /* WAW: a load into the MFMA's destination */
            ".if %c5 == 4\n\tv_mov_b32 v28, 0\n\t.endif\n\t"                                  /* WAW by VALU */
            ".if %c5 == 1\n\tds_read_b32 v[24 + %c3], v18\n\ts_waitcnt lgkmcnt(0)\n\t.endif\n\t"
            ".if %c5 == 0\n\tv_mov_b32 v[24 + %c3], 0\n\t.endif\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v28"
            : "=v"(res) : "v"(lane), "n"(K32), "n"(W), "n"(D), "n"(LDS), "n"(N)
            : "memory", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        out[lane] = res;
    } else if (BUSY) {
        // other waves (wave 4 shares SIMD 0 with wave 0) keep the matrix pipe busy
        asm volatile("v_mov_b32 v20, 0x3f803f80\n\tv_mov_b32 v21, 0x3f803f80\n\tv_mov_b32 v22, 0x3f803f80\n\tv_mov_b32 v23, 0x3f803f80\n\t"
                     "v_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\ts_nop 3\n\t"
                     ".rept 400\n\tv_mfma_f32_16x16x32_bf16 v[28:31], v[20:23], v[20:23], v[28:31]\n\t.endr\n\ts_nop 15"
                     ::: "v20", "v21", "v22", "v23", "v28", "v29", "v30", "v31");
    }
}

template <int K32, int W, int D, int LDS, int N = 1, int BUSY = 0>
float run(float* d, int threads) {
    hipMemset(d, 0, 1024);
    hipLaunchKernelGGL((k<K32, W, D, LDS, N, BUSY>), dim3(1), dim3(threads), 4096, 0, d);
    float h[64];
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    return h[0];
}

template <int K32, int W, int LDS>
void sweep(float* d, int threads) {
    const float full = K32 ? 32.f : 16.f;
    printf("%s, overwrite B dword %d by %s, %d waves: acc (full %g) at wait states 0..8,12:", K32 ? "16x16x32" : "16x16x16", W, LDS ? "ds_read" : "v_mov  ", threads / 64, full);
    printf(" %g %g %g %g %g %g %g %g %g %g\n", run<K32, W, 0, LDS>(d, threads), run<K32, W, 1, LDS>(d, threads), run<K32, W, 2, LDS>(d, threads),
           run<K32, W, 3, LDS>(d, threads), run<K32, W, 4, LDS>(d, threads), run<K32, W, 5, LDS>(d, threads), run<K32, W, 6, LDS>(d, threads),
           run<K32, W, 7, LDS>(d, threads), run<K32, W, 8, LDS>(d, threads), run<K32, W, 12, LDS>(d, threads));
}

template <int N, int LDS, int BUSY>
void chain(float* d, int threads) {
    printf("chain of %2d 16x16x32, whole B (or dword 2) overwritten by %s, %d waves, busy %d: acc (full %d) at wait states 0,1,2,4,8:", N, LDS == 2 ? "ds_read_b128" : "v_mov dword 2", threads / 64, BUSY, 32 * N);
    printf(" %g %g %g %g %g\n", run<1, 2, 0, LDS, N, BUSY>(d, threads), run<1, 2, 1, LDS, N, BUSY>(d, threads), run<1, 2, 2, LDS, N, BUSY>(d, threads),
           run<1, 2, 4, LDS, N, BUSY>(d, threads), run<1, 2, 8, LDS, N, BUSY>(d, threads));
}

template <int N, int KIND, int BUSY>
void waw(float* d, int threads) {
    printf("WAW: chain of %2d, then %s into the destination (zeros), %d waves, busy %d: final D (0 = program order) at wait states 0,1,2,4,8:", N, KIND == 3 ? "ds_read" : "v_mov  ", threads / 64, BUSY);
    printf(" %g %g %g %g %g\n", run<1, 0, 0, KIND, N, BUSY>(d, threads), run<1, 0, 1, KIND, N, BUSY>(d, threads), run<1, 0, 2, KIND, N, BUSY>(d, threads),
           run<1, 0, 4, KIND, N, BUSY>(d, threads), run<1, 0, 8, KIND, N, BUSY>(d, threads));
}

int main() {
    float* d; hipMalloc(&d, 1024);
    for (int threads : {64, 320, 576}) {
        waw<1, 3, 0>(d, threads); waw<6, 3, 0>(d, threads); waw<1, 3, 1>(d, threads); waw<2, 3, 1>(d, threads); waw<6, 3, 1>(d, threads); waw<12, 3, 1>(d, threads);
        waw<1, 4, 0>(d, threads); waw<1, 4, 1>(d, threads); waw<6, 4, 1>(d, threads);
    }
    return 0;
    for (int threads : {64, 320, 576}) {
        chain<1, 2, 0>(d, threads); chain<2, 2, 0>(d, threads); chain<6, 2, 0>(d, threads); chain<12, 2, 0>(d, threads);
        chain<6, 0, 0>(d, threads); chain<12, 0, 0>(d, threads);
        chain<1, 2, 1>(d, threads); chain<6, 2, 1>(d, threads); chain<12, 2, 1>(d, threads); chain<6, 0, 1>(d, threads); chain<12, 0, 1>(d, threads);
    }
    for (int threads : {64, 320}) {
        sweep<1, 0, 0>(d, threads); sweep<1, 1, 0>(d, threads); sweep<1, 2, 0>(d, threads); sweep<1, 3, 0>(d, threads);
        sweep<1, 0, 1>(d, threads); sweep<1, 2, 1>(d, threads); sweep<1, 3, 1>(d, threads);
        sweep<0, 0, 0>(d, threads); sweep<0, 1, 0>(d, threads); sweep<0, 1, 1>(d, threads);
    }
    return 0;
}
