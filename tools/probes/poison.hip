// Leaves a recognisable pattern in the LDS and in the vector registers of every CU, so that a kernel that reads LDS bytes or registers
// it never wrote shows it (NaNs / huge values) instead of silently re-using what an earlier launch of the same kernel left there.
// Shared object for ctypes: poison_all(stream) launches 2 x 256 workgroups x 1024 threads with 64 KB of LDS and 128 live VGPRs.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(1024) void k_poison(float* sink, uint32_t pat) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds[i] = pat;
    __syncthreads();
    uint32_t v[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) v[i] = pat + i * 0;           // fill registers with the pattern
#pragma unroll
    for (int i = 0; i < 96; ++i) asm volatile("" : "+v"(v[i]));
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 96; ++i) acc ^= v[i];
    if (acc == 0x12345u) sink[threadIdx.x] = (float)lds[threadIdx.x];
}
extern "C" int poison_all(void* stream, float* sink, uint32_t pat) {
    static bool once = [] { return hipFuncSetAttribute((const void*)k_poison, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
    (void)once;
    hipLaunchKernelGGL(k_poison, dim3(1024), dim3(1024), 160 * 1024, (hipStream_t)stream, sink, pat);
    return (int)hipGetLastError();
}
