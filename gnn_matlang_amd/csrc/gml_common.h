// Shared helpers for the gfx950 kernels of libgml_hip.so (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "gml.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define GML_WAVE 64
// group record of gml_csr_group_info: {first edge, #edges, first column, window width}; 128-row groups (bf16x3 backward)
// append one byte per lane position = the row of the group handled by that position (rows sorted by degree so a
// 16-row tile has near-equal trip counts)
#define GML_GREC_INTS(group_rows) ((group_rows) == 128 ? 4 + 128 / 4 : ((group_rows) == GML_GROUPS64_RANKED ? 4 + 64 / 4 : 4))
#define GML_NUM_CU 256
#define GML_NUM_XCD 8

// XOR key of the 16-byte chunks of a 64-byte-row bf16 image ([32 rows][32 k], hi or lo) whose MFMA fragments are read
// with ds_read_b128: lane (r16, kq) reads chunk kq ^ key(row) of row 16*blk + r16 (or of the permuted rows of the
// backward's Z projection).  ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (not
// in contiguous 16-lane groups), so the key must separate rows r and r + 4 of DIFFERENT kq: key = -(row >> 2) mod 4 is
// conflict-free for both row orders (tools/lds_sim.py; the round-1 keys (row >> 2) / (row >> 3) were 2-way).
__host__ __device__ __forceinline__ int gml_wkey(int row) { return (-(row >> 2)) & 3; }

typedef short gml_s16x4 __attribute__((ext_vector_type(4)));
typedef short gml_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) gml_s16x4 gml_lds_s16x4;

// one MFMA operand (8 k-slots) from two transposing reads: lane (t, g) of a 16-lane group receives, for j = 0..3, element
// (t & 3) of the 8-byte chunk whose address lane 4 j + (t >> 2) of the same group passed (probed: tools/probes/probe_tr.hip)
__device__ __forceinline__ bf16x8 gml_tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const gml_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gml_lds_s16x4*)(p0));
    const gml_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gml_lds_s16x4*)(p1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// return code of the last launch on this thread (no sync).  hipGetLastError: reads AND clears the thread's last-error word,
// so a failed launch is reported once, by the call that caused it -- with Peek (r02) a non-sticky error left behind by one
// failed launch, or by any third-party call, made every later gml_* call return that stale code although its own launch
// succeeded.  Sticky (device-fault) errors survive the read and keep failing every call, as they must.
static inline int gml_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GML_OK : (int)e;
}

// Opt a kernel into more than 64 KiB of dynamic LDS.  The attribute belongs to the (kernel, device) pair, so it is set
// once per device the calling process launches on (bit d of a per-call-site mask = done on device d), not once per
// process: a process that drives several devices gets it on each.
static inline hipError_t gml_allow_big_lds_impl(std::atomic<uint64_t>& done, const void* kernel, int nbytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, nbytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}
// declares `const hipError_t rcvar`; one mask per expansion (and per template instantiation it expands in)
#define GML_ALLOW_BIG_LDS(rcvar, kernel_ptr, nbytes)                                                   \
    hipError_t rcvar##_v = hipSuccess;                                                                 \
    {                                                                                                  \
        static std::atomic<uint64_t> done_{0};                                                         \
        rcvar##_v = gml_allow_big_lds_impl(done_, reinterpret_cast<const void*>(kernel_ptr), (nbytes)); \
    }                                                                                                  \
    const hipError_t rcvar = rcvar##_v;

static inline int64_t gml_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Zero fill / copy as KERNELS, not hipMemsetAsync / hipMemcpyAsync: on this ROCm the runtime calls issued while a stream
// is being captured into a HIP graph ran at capture time only -- a replayed index build then accumulated its row
// histogram on top of the previous replay's row pointers and the slot kernel wrote out of bounds (r02, bench.py's captured
// epoch).  A kernel launch is captured like every other launch of the library.
static __global__ void gml_k_fill_zero_u32(uint32_t* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
static __global__ void gml_k_copy_u32(const uint32_t* __restrict__ a, uint32_t* __restrict__ b, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
static inline void gml_zero_async(void* p, size_t nbytes, hipStream_t st) {        // nbytes: a multiple of 4
    const int64_t n = (int64_t)(nbytes / 4);
    if (n > 0) hipLaunchKernelGGL(gml_k_fill_zero_u32, dim3((unsigned)gml_cdiv(n, 256)), dim3(256), 0, st, (uint32_t*)p, n);
}
static inline void gml_copy_async(void* dst, const void* src, size_t nbytes, hipStream_t st) {
    const int64_t n = (int64_t)(nbytes / 4);
    if (n > 0) hipLaunchKernelGGL(gml_k_copy_u32, dim3((unsigned)gml_cdiv(n, 256)), dim3(256), 0, st, (const uint32_t*)src, (uint32_t*)dst, n);
}

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

// tanh for the Hadamard branch and the exact-mode edge kernels.  |x| >= 1/4: tanh(x) = 1 - 2 / (2^(2 log2(e) x) + 1) (v_exp_f32,
// v_rcp_f32; branch-free, saturates correctly at +-inf): absolute error <= ~2e-7, i.e. <= 8e-7 relative there.  |x| < 1/4: the odd
// series x + x^3 (c1 + c2 x^2 + c3 x^4 + c4 x^6) (truncation 2e-9): RELATIVE accuracy ~1e-7 down to zero.  Rounds 1-4 used the first
// form everywhere -- tanh(1e-3) came back with 2e-4 relative error -- and that, not the split products, was the floor of the forward
// error (4e-6 rms of the activations in BOTH arithmetic modes; the reference's fp32 on the CPU: 1e-7; tools/parity_diag.py).  The
// matrix-core edge chains (gml_edge_chain*_impl.h), whose issue slots are the step's second largest cost, keep the short form.
__device__ __forceinline__ float gml_tanh_short(float x) {   // the short form alone: ~2e-7 ABSOLUTE (the 8-wave forward's Hadamard columns,
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // whose argument is a bf16x3 product anyway)
    return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}
#ifndef GML_TANH_SHORT
__device__ __forceinline__ float gml_tanh_small(float x) {
    const float x2 = x * x;
    const float p = fmaf(x2, fmaf(x2, fmaf(x2, 0.021869488536155203f, -0.053968253968253971f), 0.13333333333333333f), -0.33333333333333331f);
    return fmaf(x * x2, p, x);
}
__device__ __forceinline__ float gml_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    const float tb = fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
    return fabsf(x) < 0.25f ? gml_tanh_small(x) : tb;
}

// tanh(x) and its derivative 1 - tanh(x)^2 without the cancellation of 1 - t * t near saturation (t = 1 - 1.7e-6 already loses
// 4 % there in fp32): with e = exp(2x), r = 1 / (e + 1):  t = 1 - 2 r,  1 - t^2 = 4 e r^2.  |x| is clamped to 40 (tanh = +-1 to the
// last bit far earlier; keeps e finite).  |x| < 1/4: the series value and 1 - t^2 (no cancellation there).
__device__ __forceinline__ void gml_tanh_d(float x, float& t, float& d) {
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x, -40.f, 40.f) * 2.8853900817779268f);
    const float r = __builtin_amdgcn_rcpf(e + 1.f);
    const float ts = gml_tanh_small(x);
    const bool small = fabsf(x) < 0.25f;
    t = small ? ts : fmaf(-2.f, r, 1.f);
    d = small ? fmaf(-ts, ts, 1.f) : 4.f * (e * r) * r;
}
#else
__device__ __forceinline__ float gml_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}
__device__ __forceinline__ void gml_tanh_d(float x, float& t, float& d) {
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x, -40.f, 40.f) * 2.8853900817779268f);
    const float r = __builtin_amdgcn_rcpf(e + 1.f);
    t = fmaf(-2.f, r, 1.f);
    d = 4.f * (e * r) * r;
}
#endif

// fp32 -> (hi, lo) bf16 pair with hi + lo = x to ~2^-17 relative (round-to-nearest both times).  Products of
// two such splits, a_hi b_hi + a_hi b_lo + a_lo b_hi accumulated in fp32 by the bf16 matrix cores, carry a
// relative error of ~2^-16 per term (the dropped a_lo b_lo and the lo roundings): fp32-class for the
// K <= 768 contractions of this layer, at 1/5 of the f32 matrix-pipe time and -- unlike the f32-input MFMA,
// which executes on the fp32 vector ALUs -- concurrently with the VALU work of the other waves.
__device__ __forceinline__ void gml_split2(float x0, float x1, bf16x2& hi, bf16x2& lo) {
    const f32x2 v = f32x2{x0, x1};
    hi = __builtin_convertvector(v, bf16x2);
    const f32x2 r = v - __builtin_convertvector(hi, f32x2);
    lo = __builtin_convertvector(r, bf16x2);
}

__device__ __forceinline__ void gml_split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16x2 h, l;
        gml_split2(x[2 * i], x[2 * i + 1], h, l);
        hi[2 * i] = h[0]; hi[2 * i + 1] = h[1];
        lo[2 * i] = l[0]; lo[2 * i + 1] = l[1];
    }
}

// ---- "f16x3": the same three split products on f16 pieces (11 + 11 significant bits: hi + lo = x to 2^-24 -- fp32's own rounding --
// where the bf16 pair stops at 2^-17), same MFMA rate, same operand layouts.  f16 has 5 exponent bits, so every operand goes through a
// power-of-two scale (exact) that puts its tile's / column's largest magnitude just below 2^15; elements more than 2^17 below that
// maximum lose relative (not absolute) precision gracefully: the absolute floor is 2^-39 of the maximum.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// scale s = 2^k with m * s in [2^14, 2^15) for a maximum m >= 0 (bits), and 1 / s; both kept inside the normal range
__device__ __forceinline__ void gml_f16_scale_bits(uint32_t mbits, float& s, float& inv) {
    const int e = (int)(mbits >> 23) & 0xff;
    const int sb = min(268 - e, 253);
    s = __uint_as_float((uint32_t)sb << 23);
    inv = __uint_as_float((uint32_t)(254 - sb) << 23);
}
// maximum over the wave of a non-negative float's bits (non-negative floats order like their bit patterns); all 64 lanes active
__device__ __forceinline__ uint32_t gml_wave_max_bits(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));    // row_half_mirror
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));    // row_mirror
    const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
// x * s -> (hi, lo) f16 pairs
__device__ __forceinline__ void gml_split8_f16(const float (&x)[8], float s, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 v = f32x2{x[2 * i], x[2 * i + 1]} * s;
        const f16x2 h = __builtin_convertvector(v, f16x2);
        const f32x2 r = v - __builtin_convertvector(h, f32x2);
        const f16x2 l = __builtin_convertvector(r, f16x2);
        hi[2 * i] = h[0]; hi[2 * i + 1] = h[1];
        lo[2 * i] = l[0]; lo[2 * i + 1] = l[1];
    }
}

// operand piece type of the projection: bf16 pairs (bf16x3) or f16 pairs under power-of-two scales (f16x3, gml_common.h)
template <bool F16> struct GmlPiece { using T = bf16x8; };
template <> struct GmlPiece<true> { using T = f16x8; };
__device__ __forceinline__ f32x4 gml_mfma_piece(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 gml_mfma_piece(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// sum of partial[w * n + j] over w = wl, wl + 16, ... < nparts in ascending order (the fixed order of every
// partial fold), eight clamped, unconditional loads in flight per step instead of one dependent load per add
__device__ __forceinline__ float gml_fold_column(const float* __restrict__ partial, int64_t nparts, int64_t n,
                                                 int64_t j, int wl) {
    float a = 0.f;
    for (int64_t w = wl; w < nparts; w += 16 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t w2 = w + 16 * u;
            v[u] = partial[(w2 < nparts ? w2 : nparts - 1) * n + j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (w + 16 * u < nparts) a += v[u];
    }
    return a;
}
