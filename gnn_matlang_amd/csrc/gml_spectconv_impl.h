// Fused multi-support spectral convolution for gfx950:
//
//   out[r, :] (op)= act( sum_s ( sum_{k in row r} val[pos(k), s] * x[col[k], :] ) @ W[s] + bias )
//
// replaces, in one launch, the S x (index_select, mul, scatter_add, matmul, add) of
// /root/reference/libs/spect_conv.py:76-80 + bias :93-94 (+ the relu of ML3Layer :209).
//
// Work decomposition (wave64, one 16-row tile per wave, 4 waves per workgroup):
//   lane l = (r16 = l & 15, kq = l >> 4) owns output-row r16 of the tile and, inside a chunk of
//   CH = 4*FPL input features, the FPL consecutive features [kq*FPL, kq*FPL+FPL).
//   * aggregation (VALU): the lane walks the CSR row in order (= the reference's per-target
//     summation order) and keeps acc[s][j] = H[r16][s][f] in registers - H never goes to HBM;
//   * projection (MFMA): acc[s][j] IS the A fragment of v_mfma_f32_16x16x4_f32
//     (A[i = l&15][k = l>>4]); the K order of the contraction is permuted to
//     (chunk, s, j, kq) and W is laid out in LDS once per workgroup in exactly that order, one
//     dword per lane per MFMA, conflict-free: Wl[(s*FPL + j)*NB + nb][lane];
//   * epilogue: D[row = 4*(l>>4) + reg][col = l&15] (+ old out) + bias, relu, 64-B row segments.
// f32-in MFMA is bit-identical to an fmaf chain (exact fp32), so parity is fp32-roundoff class.
#pragma once
#include "gml_common.h"

struct GmlFwdParams {
    const int32_t* rowptr;
    const int32_t* col;
    const int32_t* epos;
    const float* val;
    const float* x;
    int64_t ldx;
    const float* w;
    int64_t w_ss, w_si, w_so;
    const float* bias;
    float* out;
    int64_t ldo;
    int64_t nrows;
    int32_t S, Fin, Fout;
    uint32_t flags;
    int32_t s0;          // first support handled by this launch
    int32_t npass;       // supports handled = npass * SC
    int32_t nchunks;     // ceil(Fin / (4*FPL))
    int32_t ntiles;      // ceil(nrows / 16)
    int32_t tiles_per_wg;
    int32_t allw;        // whole W of this launch resident in LDS
    int32_t val_vec;     // value rows (S floats apart, starting at s0) keep the SC alignment class
};

template <int SC, int FPL, int NB>
__device__ __forceinline__ void gml_stage_w(float* __restrict__ dst, const GmlFwdParams& p, int pass, int c) {
    constexpr int CH = 4 * FPL;
    constexpr int WBLK = SC * FPL * NB * 64;
    for (int e = threadIdx.x; e < WBLK; e += blockDim.x) {
        const int lane = e & 63;
        int rest = e >> 6;
        const int nb = rest % NB; rest /= NB;
        const int j = rest % FPL;
        const int s = rest / FPL;
        const int f = c * CH + (lane >> 4) * FPL + j;
        const int o = nb * 16 + (lane & 15);
        float v = 0.f;
        if (f < p.Fin && o < p.Fout)
            v = p.w[(int64_t)(p.s0 + pass * SC + s) * p.w_ss + (int64_t)f * p.w_si + (int64_t)o * p.w_so];
        dst[e] = v;
    }
}

template <int SC, int FPL, int NB, bool XVEC>
__global__ __launch_bounds__(256) void gml_k_spectconv_fwd(const GmlFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];
    constexpr int CH = 4 * FPL;
    constexpr int WBLK = SC * FPL * NB * 64;
    // alignment class of a value row start (in floats): rows are S floats apart, chunk starts at s0 + pass*SC
    constexpr int VAL_ALIGN = (SC % 4 == 0) ? 4 : ((SC % 2 == 0) ? 2 : 1);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int wg = gml_xcd_remap(blockIdx.x, gridDim.x);
    const int t0 = wg * p.tiles_per_wg;
    const int t1 = min(t0 + p.tiles_per_wg, p.ntiles);

    if (p.allw) {
        for (int pass = 0; pass < p.npass; ++pass)
            for (int c = 0; c < p.nchunks; ++c)
                gml_stage_w<SC, FPL, NB>(lds_w + (pass * p.nchunks + c) * WBLK, p, pass, c);
        __syncthreads();
    }

    for (int tb = t0; tb < t1; tb += 4) {
        const int tile = tb + wave;
        const int64_t row = (int64_t)tile * 16 + r16;
        const bool rvalid = tile < t1 && row < p.nrows;
        const int kbeg = rvalid ? p.rowptr[row] : 0;
        const int kend = rvalid ? p.rowptr[row + 1] : 0;

        f32x4 oacc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int pass = 0; pass < p.npass; ++pass) {
            const int sbase = p.s0 + pass * SC;
            for (int c = 0; c < p.nchunks; ++c) {
                const float* wl;
                if (p.allw) {
                    wl = lds_w + (pass * p.nchunks + c) * WBLK;
                } else {
                    __syncthreads();
                    gml_stage_w<SC, FPL, NB>(lds_w, p, pass, c);
                    __syncthreads();
                    wl = lds_w;
                }
                float acc[SC][FPL];
#pragma unroll
                for (int s = 0; s < SC; ++s)
#pragma unroll
                    for (int j = 0; j < FPL; ++j) acc[s][j] = 0.f;

                const int f0 = c * CH + kq * FPL;
                for (int k = kbeg; k < kend; ++k) {
                    const int src = p.col[k];
                    const int64_t pk = p.epos ? (int64_t)p.epos[k] : (int64_t)k;
                    float ev[SC];
                    if (p.val_vec) gml_load_row<SC, VAL_ALIGN>(p.val + pk * p.S + sbase, ev);
                    else gml_load_row<SC, 1>(p.val + pk * p.S + sbase, ev);
                    float xv[FPL];
                    const float* xr = p.x + (int64_t)src * p.ldx + f0;
                    if constexpr (XVEC) {
#pragma unroll
                        for (int q = 0; q < FPL / 4; ++q) {
                            f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (f0 + 4 * q < p.Fin) t = *reinterpret_cast<const f32x4*>(xr + 4 * q);
                            xv[4 * q] = t.x; xv[4 * q + 1] = t.y; xv[4 * q + 2] = t.z; xv[4 * q + 3] = t.w;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < FPL; ++j) xv[j] = (f0 + j < p.Fin) ? xr[j] : 0.f;
                    }
#pragma unroll
                    for (int s = 0; s < SC; ++s)
#pragma unroll
                        for (int j = 0; j < FPL; ++j) acc[s][j] = fmaf(ev[s], xv[j], acc[s][j]);
                }

#pragma unroll
                for (int s = 0; s < SC; ++s)
#pragma unroll
                    for (int j = 0; j < FPL; ++j) {
                        const float a = acc[s][j];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const float b = wl[((s * FPL + j) * NB + nb) * 64 + lane];
                            oacc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, oacc[nb], 0, 0, 0);
                        }
                    }
            }
        }

        if (tile < t1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int o = nb * 16 + r16;
                if (o < p.Fout) {
                    const float bv = p.bias ? p.bias[o] : 0.f;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int64_t orow = (int64_t)tile * 16 + 4 * kq + reg;
                        if (orow < p.nrows) {
                            float* dst = p.out + orow * p.ldo + o;
                            float v = oacc[nb][reg] + bv;
                            if (p.flags & GML_ACCUM) v += *dst;
                            if (p.flags & GML_RELU) v = fmaxf(v, 0.f);
                            *dst = v;
                        }
                    }
                }
            }
        }
    }
}

// one (SC, FPL) family: NB in {1,2,4,8}, XVEC in {0,1}
template <int SC, int FPL>
int gml_launch_fwd_family(const GmlFwdParams& p, int NB, bool xvec, dim3 grid, size_t lds, hipStream_t st);

#define GML_FWD_CASE(NBV, XV)                                                                         \
    hipLaunchKernelGGL((gml_k_spectconv_fwd<SC, FPL, NBV, XV>), grid, dim3(256), lds, st, p);          \
    return gml_launch_status();

#define GML_DEFINE_FWD_FAMILY(SCV, FPLV)                                                              \
    template <>                                                                                       \
    int gml_launch_fwd_family<SCV, FPLV>(const GmlFwdParams& p, int NB, bool xvec, dim3 grid,         \
                                         size_t lds, hipStream_t st) {                                \
        constexpr int SC = SCV, FPL = FPLV;                                                           \
        if (xvec) {                                                                                   \
            switch (NB) {                                                                             \
                case 1: { GML_FWD_CASE(1, true) }                                                     \
                case 2: { GML_FWD_CASE(2, true) }                                                     \
                case 4: { GML_FWD_CASE(4, true) }                                                     \
                case 8: { GML_FWD_CASE(8, true) }                                                     \
            }                                                                                         \
        } else {                                                                                      \
            switch (NB) {                                                                             \
                case 1: { GML_FWD_CASE(1, false) }                                                    \
                case 2: { GML_FWD_CASE(2, false) }                                                    \
                case 4: { GML_FWD_CASE(4, false) }                                                    \
                case 8: { GML_FWD_CASE(8, false) }                                                    \
            }                                                                                         \
        }                                                                                             \
        return GML_E_UNSUPPORTED;                                                                     \
    }
