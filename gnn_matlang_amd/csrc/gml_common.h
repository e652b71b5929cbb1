// Shared helpers for the gfx950 kernels of libgml_hip.so (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gml.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GML_WAVE 64
#define GML_NUM_CU 256
#define GML_NUM_XCD 8

// return code of the last launch on this thread (no sync)
static inline int gml_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GML_OK : (int)e;
}

static inline int64_t gml_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed, speed only).
// Give each XCD one contiguous range of work items so neighbouring row tiles (which gather the
// same X rows and value lines) share an L2.  Bijective for any grid size.
__device__ __forceinline__ int gml_xcd_remap(int b, int nblk) {
    const int q = nblk / GML_NUM_XCD, r = nblk % GML_NUM_XCD;
    const int xcd = b % GML_NUM_XCD, idx = b / GML_NUM_XCD;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// load N consecutive floats, using the widest access the static alignment class allows
template <int N, int ALIGN_FLOATS>
__device__ __forceinline__ void gml_load_row(const float* __restrict__ p, float (&v)[N]) {
    if constexpr (ALIGN_FLOATS >= 4 && N % 4 == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * i);
            v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    } else if constexpr (ALIGN_FLOATS >= 2 && N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(p + 2 * i);
            v[2 * i] = t.x; v[2 * i + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = p[i];
    }
}

// tanh for the edge / Hadamard branches: tanh(x) = 1 - 2 / (2^(2 log2(e) x) + 1), five instructions
// (v_exp_f32, v_rcp_f32), branch-free, saturates correctly at +-inf.  Absolute error <= ~2e-7 (a few
// ulp of 1.0); near 0 that is absolute, not relative, accuracy -- fp32-roundoff class for the sums and
// products the layer forms from it (parity tolerance: 1e-4 of the tensor's max).
__device__ __forceinline__ float gml_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}
