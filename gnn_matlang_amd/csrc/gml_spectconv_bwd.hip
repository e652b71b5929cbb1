// C-ABI dispatch of the fused backward kernel + the deterministic partial fold.
#include "gml_spectconv_bwd2_impl.h"
#include "gml_spectconv_bwd3_impl.h"
#include "gml_spectconv_bwd4_impl.h"
#include "gml_spectconv_bwd5_impl.h"
#include <stdlib.h>

__global__ __launch_bounds__(256) void gml_k_reduce_rows(const float* __restrict__ partial, int64_t nparts, int64_t n,
                                                        float* __restrict__ out) {
    __shared__ float red[16][17];
    const int jl = threadIdx.x & 15, wl = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 16 + jl;
    float a = 0.f;
    if (j < n)
        for (int64_t w = wl; w < nparts; w += 16) a += partial[w * n + j];
    red[wl][jl] = a;
    __syncthreads();
    if (wl == 0 && j < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][jl];
        out[j] = t;
    }
}

#define GML_DECL_BWD(S, A, B) \
    template <> int gml_launch_bwd<S, A, B>(const GmlBwdParams&, dim3, size_t, hipStream_t);
GML_DECL_BWD(8, 2, 2) GML_DECL_BWD(8, 1, 2) GML_DECL_BWD(4, 2, 2) GML_DECL_BWD(4, 1, 2)
GML_DECL_BWD(12, 2, 1) GML_DECL_BWD(12, 1, 1) GML_DECL_BWD(6, 3, 2) GML_DECL_BWD(6, 1, 2)
GML_DECL_BWD(4, 3, 2) GML_DECL_BWD(6, 2, 2) GML_DECL_BWD(8, 2, 1) GML_DECL_BWD(4, 4, 2)

#ifdef GML_BWD2_TIMING
static unsigned long long* bwd2_prof_buf();
#endif
#define GML_DECL_BWD2(S, A) template <> int gml_launch_bwd2<S, A>(const GmlBwdParams&, dim3, size_t, hipStream_t);
GML_DECL_BWD2(8, 2) GML_DECL_BWD2(8, 1) GML_DECL_BWD2(6, 2) GML_DECL_BWD2(6, 1)
GML_DECL_BWD2(4, 2) GML_DECL_BWD2(4, 1) GML_DECL_BWD2(2, 2) GML_DECL_BWD2(2, 1)

#define GML_DECL_BWD3(S, A, W) template <> int gml_launch_bwd3<S, A, W>(const GmlBwdParams&, dim3, size_t, hipStream_t);
GML_DECL_BWD3(8, 2, 8) GML_DECL_BWD3(8, 1, 8) GML_DECL_BWD3(6, 2, 8) GML_DECL_BWD3(6, 1, 8)
GML_DECL_BWD3(4, 2, 8) GML_DECL_BWD3(4, 1, 8) GML_DECL_BWD3(2, 2, 8) GML_DECL_BWD3(2, 1, 8)
GML_DECL_BWD3(8, 2, 4) GML_DECL_BWD3(8, 1, 4)
GML_DECL_BWD3(6, 3, 8) GML_DECL_BWD3(4, 3, 8)             /* 33 .. 48 input features in one launch (gml_bwd3_fam_g.hip) */
template <> int gml_launch_bwd3<12, 2, 8, 1>(const GmlBwdParams&, dim3, size_t, hipStream_t);   /* counting.py: S = 12, Fout <= 16 */
template <> int gml_launch_bwd3<12, 1, 8, 1>(const GmlBwdParams&, dim3, size_t, hipStream_t);
template <> int gml_launch_bwd3<8, 2, 8, 1>(const GmlBwdParams&, dim3, size_t, hipStream_t);    /* S = 8, Fout <= 16 (gml_bwd3_fam_f.hip) */
template <> int gml_launch_bwd3<8, 1, 8, 1>(const GmlBwdParams&, dim3, size_t, hipStream_t);

#define GML_DECL_BWD4(S, A) template <> int gml_launch_bwd4<S, A>(const GmlBwdParams&, dim3, hipStream_t);
GML_DECL_BWD4(8, 2) GML_DECL_BWD4(8, 1) GML_DECL_BWD4(4, 2) GML_DECL_BWD4(4, 1)
/* bwd4 (LDS-DMA landing ring) runs when the caller sets GML_DMA_RING in flags or the process was started with GML_BWD_DMA=1.
   It is NOT the default: at ZINC shapes it measures 3-4 % slower than bwd3 (profiles/r03_bwd4_vs_bwd3_phases.txt: what the ring
   saves -- commit and the barriers around it, 4 % -- the two-supports-per-slab dW phase it forces gives back). */
template <> int gml_launch_bwd5<2>(const GmlBwdParams&, dim3, size_t, hipStream_t);
/* bwd5 (12 waves: two edge passes over 64 accumulators + four helper waves, VERDICT r04 item 1's form) for the ZINC shape class.
   OPT-IN (GML_BWD5=1 in the environment): parity-green, but 8-23 % SLOWER than bwd3 in every schedule tried
   (profiles/r05_bwd5_ab.txt) -- the second edge pass costs more than the third wave per SIMD and the off-loaded phases give back */
static bool bwd5_env() { static const bool v = [] { const char* e = getenv("GML_BWD5"); return e && e[0] == '1'; }(); return v; }
static bool bwd4_env() { static const bool v = [] { const char* e = getenv("GML_BWD_DMA"); return e && e[0] == '1'; }(); return v; }

struct BwdPlan {
    int ok, S, nfb, nob, grid, groups_per_wg, ecap, xcap, rows;   /* rows: 64 (f32 MFMA kernel) or 128 / 64 (bf16x3 kernels) */
    int layout, nw;                                                /* bf16x3: layout 2 (bwd2) or 3 (bwd3), waves per workgroup */
    size_t lds;
};

/* bf16x3 kernel selection (experiments: GML_BWD_LAYOUT = 2 | 3, GML_BWD_NW = 4 | 8), read once */
static int bwd_layout_env() { static const int v = [] { const char* e = getenv("GML_BWD_LAYOUT"); return e ? atoi(e) : 3; }(); return v; }
static int bwd_nw_env() { static const int v = [] { const char* e = getenv("GML_BWD_NW"); return e ? atoi(e) : 8; }(); return v; }
static int bwd3_nw(int S) { return (bwd_nw_env() == 4 && S == 8) ? 4 : 8; }

#define GML_BWD_TRY(SV, A, B)                                                              \
    if (S == SV && nfb == A && nob == B) {                                                 \
        pl.lds = GmlBwdCfg<SV, A, B>::lds_bytes(pl.ecap, pl.xcap);                         \
        pl.ok = pl.lds <= 160 * 1024;                                                      \
        return pl;                                                                         \
    }

static bool bwd2_shape(int S, int Fin, int Fout, uint32_t flags) {
    return !(flags & GML_F32_MFMA) && (S == 2 || S == 4 || S == 6 || S == 8) && Fin <= 32 && Fout > 16 && Fout <= 32;
}

/* 33 .. 48 input features on the 8-wave kernel in ONE launch (NFB = 3; GML_BWD_WIDE48=0 in the environment: off, for A/B against the
   two feature-slice launches the host otherwise makes) */
static bool bwd48_env() { static const bool v = [] { const char* e = getenv("GML_BWD_WIDE48"); return !(e && e[0] == '0'); }(); return v; }
static bool bwd3_wide_shape(int S, int Fin, int Fout, uint32_t flags) {
    return !(flags & GML_F32_MFMA) && (S == 4 || S == 6) && Fin > 32 && Fin <= 48 && Fout > 16 && Fout <= 32 && bwd48_env();
}

/* shapes of the third layout alone: counting.py's 12 supports with Fout <= 16 (gml_k_spectconv_bwd3<12, NFB, 8, XV, false, 1>) */
static bool bwd3_only_shape(int S, int Fin, int Fout, uint32_t flags) {
    return !(flags & GML_F32_MFMA) && (S == 12 || S == 8) && Fin <= 32 && Fout <= 16;
}

static BwdPlan plan_bwd(int64_t num_rows, int S, int Fin, int Fout, int max_edges, int max_window, uint32_t flags) {
    BwdPlan pl;
    pl.ok = 0; pl.S = S; pl.rows = 64; pl.layout = 0; pl.nw = 4;
    const int nfb = (Fin + 15) / 16, nob = (Fout + 15) / 16;
    pl.nfb = nfb; pl.nob = nob;
    pl.ecap = (max_edges + 15) / 16 * 16;
    if (pl.ecap < 64) pl.ecap = 64;
    pl.xcap = (max_window + 15) / 16 * 16;
    if (pl.xcap < 64) pl.xcap = 64;
    if ((bwd2_shape(S, Fin, Fout, flags) && bwd_layout_env() != 2) || bwd3_only_shape(S, Fin, Fout, flags) || bwd3_wide_shape(S, Fin, Fout, flags)) {   /* bf16x3 kernel, third layout */
        pl.layout = 3; pl.nw = (S == 8 && nob == 1) ? 8 : bwd3_nw(S); pl.rows = 16 * pl.nw;
        pl.nfb = (Fin + 15) / 16;
        const int ng = (int)gml_cdiv(num_rows, pl.rows);
        const int wgs = GML_NUM_CU * (pl.nw == 4 ? 2 : 1);
        int grid3 = ng < wgs ? ng : wgs;
        if (grid3 < 1) grid3 = 1;
        pl.groups_per_wg = (int)gml_cdiv(ng, grid3);
        pl.grid = pl.groups_per_wg > 0 ? (int)gml_cdiv(ng, pl.groups_per_wg) : 1;
#define GML_BWD3_LDS(SV, A, W) if (S == SV && pl.nfb == A && pl.nw == W) pl.lds = GmlBwd3Cfg<SV, A, W>::lds_bytes(pl.ecap, pl.xcap);
        pl.lds = 0;
        GML_BWD3_LDS(8, 2, 8) GML_BWD3_LDS(8, 1, 8) GML_BWD3_LDS(6, 2, 8) GML_BWD3_LDS(6, 1, 8)
        GML_BWD3_LDS(4, 2, 8) GML_BWD3_LDS(4, 1, 8) GML_BWD3_LDS(2, 2, 8) GML_BWD3_LDS(2, 1, 8)
        GML_BWD3_LDS(8, 2, 4) GML_BWD3_LDS(8, 1, 4) GML_BWD3_LDS(6, 3, 8) GML_BWD3_LDS(4, 3, 8)
        if (S == 12) pl.lds = pl.nfb == 2 ? GmlBwd3Cfg<12, 2, 8, 1>::lds_bytes(pl.ecap, pl.xcap) : GmlBwd3Cfg<12, 1, 8, 1>::lds_bytes(pl.ecap, pl.xcap);
        if (S == 8 && nob == 1) { pl.lds = pl.nfb == 2 ? GmlBwd3Cfg<8, 2, 8, 1>::lds_bytes(pl.ecap, pl.xcap) : GmlBwd3Cfg<8, 1, 8, 1>::lds_bytes(pl.ecap, pl.xcap); }
        if (pl.lds > 0 && S == 8) pl.lds += 512;            /* the DZ instantiation's wmix rows (gml_spectconv_bwd_mix: compiled for S = 8) */
        /* (a group too large for the LDS: not ok -- the caller then asks for the f32-MFMA kernel with ITS group records) */
        pl.ok = pl.lds > 0 && pl.lds <= (pl.nw == 4 ? 80 : 160) * 1024;
        return pl;
    } else if (bwd2_shape(S, Fin, Fout, flags)) {          /* bf16x3 kernel: 128-row groups, one 8-wave workgroup per CU */
        pl.rows = 128; pl.layout = 2; pl.nw = 8;
        pl.nfb = (Fin + 15) / 16;
        if (pl.ecap < 512) pl.ecap = 512;                  /* the value rows' region later holds the P^T slab */
        const int ng = (int)gml_cdiv(num_rows, 128);
        int grid2 = ng < GML_NUM_CU ? ng : GML_NUM_CU;
        if (grid2 < 1) grid2 = 1;
        pl.groups_per_wg = (int)gml_cdiv(ng, grid2);
        pl.grid = pl.groups_per_wg > 0 ? (int)gml_cdiv(ng, pl.groups_per_wg) : 1;
        switch (S) {
            case 8: pl.lds = GmlBwd2Cfg<8, 2>::lds_bytes(pl.ecap, pl.xcap); break;
            case 6: pl.lds = GmlBwd2Cfg<6, 2>::lds_bytes(pl.ecap, pl.xcap); break;
            case 4: pl.lds = GmlBwd2Cfg<4, 2>::lds_bytes(pl.ecap, pl.xcap); break;
            default: pl.lds = GmlBwd2Cfg<2, 2>::lds_bytes(pl.ecap, pl.xcap); break;
        }
        pl.ok = pl.lds <= 160 * 1024;
        return pl;
    }
    const int ngroups = (int)gml_cdiv(num_rows, 64);
    int grid = ngroups < GML_NUM_CU * 2 ? ngroups : GML_NUM_CU * 2;
    if (grid < 1) grid = 1;
    pl.groups_per_wg = (int)gml_cdiv(ngroups, grid);
    pl.grid = pl.groups_per_wg > 0 ? (int)gml_cdiv(ngroups, pl.groups_per_wg) : 1;
    pl.lds = 0;
    GML_BWD_TRY(8, 2, 2) GML_BWD_TRY(8, 1, 2) GML_BWD_TRY(4, 2, 2) GML_BWD_TRY(4, 1, 2)
    GML_BWD_TRY(12, 2, 1) GML_BWD_TRY(12, 1, 1) GML_BWD_TRY(6, 3, 2) GML_BWD_TRY(6, 1, 2)
    GML_BWD_TRY(4, 3, 2) GML_BWD_TRY(6, 2, 2) GML_BWD_TRY(8, 2, 1) GML_BWD_TRY(4, 4, 2)
    return pl;
}

extern "C" int gml_spectconv_bwd_group_rows(int32_t S, int32_t Fin, int32_t Fout, uint32_t flags) {
    if (S <= 0 || Fin <= 0 || Fout <= 0) return 0;
    if (bwd2_shape(S, Fin, Fout, flags)) return (bwd_layout_env() != 2 && bwd3_nw(S) == 4) ? GML_GROUPS64_RANKED : 128;
    if (bwd3_only_shape(S, Fin, Fout, flags) || bwd3_wide_shape(S, Fin, Fout, flags)) return 128;
    const BwdPlan pl = plan_bwd(64, S, Fin, Fout, 64, 64, flags | GML_F32_MFMA);
    return pl.ok ? 64 : 0;
}

extern "C" size_t gml_spectconv_bwd_workspace_bytes(int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                                                    int32_t max_group_edges, int32_t max_group_window, uint32_t flags) {
    if (num_rows <= 0 || S <= 0 || Fin <= 0 || Fout <= 0) return 0;
    const BwdPlan pl = plan_bwd(num_rows, S, Fin, Fout, max_group_edges, max_group_window, flags);
    if (!pl.ok) return 0;                                     /* 0 = this shape has no fused backward */
    return (size_t)pl.grid * S * Fin * Fout * sizeof(float);
}

static int spectconv_bwd_impl(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                              const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                              float* dx, int64_t lddx, float* dval, float* dw,
                              int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                              int32_t max_group_edges, int32_t max_group_window, uint32_t flags,
                              void* ws, size_t ws_bytes, gml_stream_t stream, const float* dz, const float* wmix, int32_t nmix,
                              int32_t relu_cols = 0, const float* wmix2 = nullptr, int32_t nmix1 = -1,
                              const float* hb11 = nullptr, const float* hb12 = nullptr, float* hpart = nullptr) {
    if (num_rows < 0 || S <= 0 || Fin <= 0 || Fout <= 0 || ldx < Fin || ldg < Fout) return GML_E_BADARG;
    if (dx && lddx < Fin) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (num_rows == 0) {
        if (dw) gml_zero_async(dw, sizeof(float) * S * Fin * Fout, st);
        return gml_launch_status();
    }
    if (!rowptr || !ginfo || !x || !g || !w) return GML_E_BADARG;
    if ((((uintptr_t)val | (uintptr_t)dval) & 15) != 0) return GML_E_BADARG;
    const BwdPlan pl = plan_bwd(num_rows, S, Fin, Fout, max_group_edges, max_group_window, flags);
    if (!pl.ok) return GML_E_UNSUPPORTED;
    const size_t need = (size_t)pl.grid * S * Fin * Fout * sizeof(float);
    if (dw && (!ws || ws_bytes < need)) return GML_E_WORKSPACE;

    GmlBwdParams p;
    p.rowptr = rowptr; p.col = col; p.ginfo = ginfo; p.val = val; p.x = x; p.ldx = ldx; p.g = g; p.ldg = ldg;
    p.w = w; p.dx = dx; p.lddx = lddx; p.dval = dval; p.dw_partial = dw ? (float*)ws : nullptr;
    p.nrows = num_rows; p.S = S; p.Fin = Fin; p.Fout = Fout; p.flags = flags;
    p.dz = dz; p.wmix = wmix; p.nmix = nmix; p.relu_cols = relu_cols;
    p.wmix2 = wmix2; p.nmix1 = (wmix2 && nmix1 >= 0 && nmix1 <= nmix) ? nmix1 : nmix;
    p.hb11 = hb11; p.hb12 = hb12; p.hpart = hpart;
    if (hpart != nullptr && (pl.layout != 3 || pl.nw != 8 || (flags & (GML_ACCUM | GML_DVAL_ACCUM)) || !dw ||
                             pl.lds + GML_BWD3_HAD_LDS(128, 8) > 160 * 1024))
        return GML_E_UNSUPPORTED;
    if (dz != nullptr && (pl.layout != 3 || (flags & GML_ACCUM) || (((uintptr_t)dz) & 15) != 0))
        return GML_E_UNSUPPORTED;
#ifdef GML_BWD2_TIMING
    p.prof = bwd2_prof_buf();
#endif
    /* float4-addressable x rows.  The 8-wave kernel (layout 3) reads whole float4 groups up to roundup4(Fin) and discards the columns
       >= Fin itself, so rows that merely HAVE those columns (ldx >= roundup4(Fin): the zero-padded [N, 28] copy of ZINC's 25 input
       features) take its vector road; the other families want Fin % 4 == 0 */
    const bool xrows4 = (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0);
    const bool xvec_strict = xrows4 && (Fin % 4 == 0);
    p.xvec = pl.layout == 3 ? (xrows4 && (Fin + 3) / 4 * 4 <= ldx) : xvec_strict;
    p.dxvec = dx && (Fin % 4 == 0) && (lddx % 4 == 0) && (((uintptr_t)dx & 15) == 0);
    /* float4 groups up to roundup4(Fout) must exist in every row: true when ldg covers them (zero padded) */
    p.gvec = (ldg % 4 == 0) && ((Fout + 3) / 4 * 4 <= ldg) && (((uintptr_t)g & 15) == 0);
    /* the 8-wave kernel prefetches the g window with float4 loads whether or not it will use them */
    if (pl.layout >= 2 && !p.gvec) return GML_E_BADARG;
    p.ngroups = (int)gml_cdiv(num_rows, pl.rows); p.groups_per_wg = pl.groups_per_wg; p.ecap = pl.ecap; p.xcap = pl.xcap;
    const int nfb = pl.nfb, nob = pl.nob;
    int rc = GML_E_UNSUPPORTED;
    /* LDS-DMA landing ring (bwd4): 8 waves, float4-addressable x / g rows, dx from dz or from zero, every group inside the
       kernel's staging capacities (+3 edges / +7 window rows of alignment slack), 32-bit row offsets */
    bool dma = pl.layout == 3 && pl.nw == 8 && (bwd4_env() || (flags & GML_DMA_RING)) && (S == 8 || S == 4) && xvec_strict && p.gvec && !(flags & GML_ACCUM) &&
               (!dx || p.dxvec || dz == nullptr) && (num_rows + 16) * (ldg > ldx ? ldg : ldx) * 4 < (int64_t)INT32_MAX;
    if (flags & GML_DVAL_ACCUM) {                            /* dval += : the 8-wave bf16x3 kernel's copy-out only */
        if (pl.layout != 3 || (S == 8 && nob == 2)) return GML_E_UNSUPPORTED;   /* (not compiled into the ZINC shape class) */
        dma = false;
    }
    if (dma) {
        const int ecap4 = S == 8 ? (pl.nfb == 2 ? GmlBwd4Cfg<8, 2>::ECAP : GmlBwd4Cfg<8, 1>::ECAP) : (pl.nfb == 2 ? GmlBwd4Cfg<4, 2>::ECAP : GmlBwd4Cfg<4, 1>::ECAP);
        dma = max_group_edges + 3 <= ecap4 && max_group_window + 7 <= GmlBwd4Cfg<8, 2>::XCAP;
    }
    const bool five = !dma && pl.layout == 3 && pl.nw == 8 && S == 8 && nob == 2 && nfb == 2 && bwd5_env() && p.gvec &&
                      !(flags & GML_DVAL_ACCUM) && GmlBwd5Cfg<2>::lds_bytes(pl.ecap, pl.xcap) <= 160 * 1024 &&
                      (dz == nullptr || (xvec_strict && p.dxvec));
    if (five && !xvec_strict) p.xvec = 0;
    if (hpart != nullptr) {                                  /* the output stage inside the 8-wave kernel (gml_spectconv_bwd_had) */
        if (five || dma) return GML_E_UNSUPPORTED;
        rc = gml_launch_bwd3_had(p, dim3(pl.grid), pl.lds + GML_BWD3_HAD_LDS(128, 8), st);
    } else if (five) {
        rc = gml_launch_bwd5<2>(p, dim3(pl.grid), GmlBwd5Cfg<2>::lds_bytes(pl.ecap, pl.xcap), st);
    } else if (dma) {
#define GML_BWD4_GO(SV, A) if (S == SV && nfb == A) rc = gml_launch_bwd4<SV, A>(p, dim3(pl.grid), st);
        GML_BWD4_GO(8, 2) GML_BWD4_GO(8, 1) GML_BWD4_GO(4, 2) GML_BWD4_GO(4, 1)
    } else if (pl.layout == 3) {
#define GML_BWD3_GO(SV, A, W) if (S == SV && nfb == A && pl.nw == W) rc = gml_launch_bwd3<SV, A, W>(p, dim3(pl.grid), pl.lds, st);
        if (S == 8 && nob == 1) rc = nfb == 2 ? gml_launch_bwd3<8, 2, 8, 1>(p, dim3(pl.grid), pl.lds, st) : gml_launch_bwd3<8, 1, 8, 1>(p, dim3(pl.grid), pl.lds, st);
        else {
        GML_BWD3_GO(8, 2, 8) GML_BWD3_GO(8, 1, 8) GML_BWD3_GO(6, 2, 8) GML_BWD3_GO(6, 1, 8)
        GML_BWD3_GO(4, 2, 8) GML_BWD3_GO(4, 1, 8) GML_BWD3_GO(2, 2, 8) GML_BWD3_GO(2, 1, 8)
        GML_BWD3_GO(8, 2, 4) GML_BWD3_GO(8, 1, 4) GML_BWD3_GO(6, 3, 8) GML_BWD3_GO(4, 3, 8)
        }
        if (S == 12) rc = nfb == 2 ? gml_launch_bwd3<12, 2, 8, 1>(p, dim3(pl.grid), pl.lds, st) : gml_launch_bwd3<12, 1, 8, 1>(p, dim3(pl.grid), pl.lds, st);
    } else if (pl.layout == 2) {
#define GML_BWD2_GO(SV, A) if (S == SV && nfb == A) rc = gml_launch_bwd2<SV, A>(p, dim3(pl.grid), pl.lds, st);
        GML_BWD2_GO(8, 2) GML_BWD2_GO(8, 1) GML_BWD2_GO(6, 2) GML_BWD2_GO(6, 1)
        GML_BWD2_GO(4, 2) GML_BWD2_GO(4, 1) GML_BWD2_GO(2, 2) GML_BWD2_GO(2, 1)
    } else {
#define GML_BWD_GO(SV, A, B) \
    if (S == SV && nfb == A && nob == B) rc = gml_launch_bwd<SV, A, B>(p, dim3(pl.grid), pl.lds, st);
    GML_BWD_GO(8, 2, 2) GML_BWD_GO(8, 1, 2) GML_BWD_GO(4, 2, 2) GML_BWD_GO(4, 1, 2)
    GML_BWD_GO(12, 2, 1) GML_BWD_GO(12, 1, 1) GML_BWD_GO(6, 3, 2) GML_BWD_GO(6, 1, 2)
    GML_BWD_GO(4, 3, 2) GML_BWD_GO(6, 2, 2) GML_BWD_GO(8, 2, 1) GML_BWD_GO(4, 4, 2)
    }
    if (rc != GML_OK) return rc;
    if (dw && !(flags & GML_NO_FOLD)) {
        const int64_t n = (int64_t)S * Fin * Fout;
        hipLaunchKernelGGL(gml_k_reduce_rows, dim3((unsigned)gml_cdiv(n, 16)), dim3(256), 0, st,
                           (const float*)ws, (int64_t)pl.grid, n, dw);
        return gml_launch_status();
    }
    return GML_OK;
}

extern "C" int gml_spectconv_bwd(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                                 const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                                 float* dx, int64_t lddx, float* dval, float* dw,
                                 int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                                 int32_t max_group_edges, int32_t max_group_window, uint32_t flags,
                                 void* ws, size_t ws_bytes, gml_stream_t stream) {
    return spectconv_bwd_impl(rowptr, col, ginfo, val, x, ldx, g, ldg, w, dx, lddx, dval, dw, num_rows, S, Fin, Fout,
                              max_group_edges, max_group_window, flags, ws, ws_bytes, stream, nullptr, nullptr, 0);
}

// 1 when gml_spectconv_bwd_mix serves this shape (the bf16x3 8-wave kernel's DZ instantiation: S = 8, 16 < Fin <= 32)
extern "C" int gml_spectconv_bwd_mix_supported(int32_t S, int32_t Fin, int32_t Fout, int32_t nmix, uint32_t flags) {
    if (nmix < 1 || nmix > 4 || S != 8 || Fin <= 16 || Fin > 32 || Fin % 4 != 0) return 0;
    return bwd2_shape(S, Fin, Fout, flags) && bwd_layout_env() != 2 && bwd3_nw(S) == 8;
}

// gml_spectconv_bwd with dx = conv part + dz wmix: dz [num_rows, 4] (contiguous, 16-byte aligned; columns >= nmix ignored),
// wmix [nmix, Fin].  The ML3Layer's Hadamard branch hands its share of dx over as the 4 pre-activation gradients per row
// (gml_ml3_split_bwd's dz output) instead of a written-then-re-read [N, Fin] array.  dx must be wanted and float4-addressable.
extern "C" int gml_spectconv_bwd_mix(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                                     const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                                     float* dx, int64_t lddx, float* dval, float* dw, const float* dz, const float* wmix,
                                     int32_t nmix, int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                                     int32_t max_group_edges, int32_t max_group_window, uint32_t flags,
                                     void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (!dz || !wmix || !dx || !gml_spectconv_bwd_mix_supported(S, Fin, Fout, nmix, flags)) return GML_E_UNSUPPORTED;
    return spectconv_bwd_impl(rowptr, col, ginfo, val, x, ldx, g, ldg, w, dx, lddx, dval, dw, num_rows, S, Fin, Fout,
                              max_group_edges, max_group_window, flags, ws, ws_bytes, stream, dz, wmix, nmix);
}

// gml_spectconv_bwd_mix for an ML3Layer whose input x is the output of another ML3Layer ([relu(conv) | Hadamard columns],
// libs/spect_conv.py:209-212, stacked as in Zinc12k.py:338-341): dx[:, f] for f < relu_cols is written multiplied by
// (x[:, f] > 0) -- the relu of the layer below applied where its gradient is produced (the kernel holds those x rows anyway).
// The layer below then runs gml_ml3_split_bwd_ex in its pre-masked form (y = G = NULL): no saved output read, no G written.
extern "C" int gml_spectconv_bwd_mix_relu(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                                          const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                                          float* dx, int64_t lddx, float* dval, float* dw, const float* dz, const float* wmix,
                                          int32_t nmix, int32_t relu_cols, int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                                          int32_t max_group_edges, int32_t max_group_window, uint32_t flags,
                                          void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (relu_cols < 0 || relu_cols > Fin) return GML_E_BADARG;
    if (!dz || !wmix || !dx || !gml_spectconv_bwd_mix_supported(S, Fin, Fout, nmix, flags)) return GML_E_UNSUPPORTED;
    if (relu_cols > 0 && (bwd4_env() || (flags & GML_DMA_RING))) return GML_E_UNSUPPORTED;   /* (the ring form has no mask) */
    return spectconv_bwd_impl(rowptr, col, ginfo, val, x, ldx, g, ldg, w, dx, lddx, dval, dw, num_rows, S, Fin, Fout,
                              max_group_edges, max_group_window, flags, ws, ws_bytes, stream, dz, wmix, nmix, relu_cols);
}

#ifdef GML_BWD2_TIMING
// debug build only: per-phase cycle sums of thread 0 of every workgroup
// (0 dW of the previous group + barrier, 1 commit, 2 Z projection, 3 edge, 4 barrier, 5 dX chain + dx stores, 6 tail,
//  8 old-dx loads + next group's loads issued, 9 dval stores, 10 P split)
static unsigned long long* bwd2_prof_buf() {
    static unsigned long long* b = [] { unsigned long long* q = nullptr; hipMalloc(&q, 128); hipMemset(q, 0, 128); return q; }();
    return b;
}
extern "C" int gml_debug_bwd2_prof(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpy(out, bwd2_prof_buf(), 128, hipMemcpyDeviceToHost);
    if (e == hipSuccess && reset) e = hipMemset(bwd2_prof_buf(), 0, 128);
    return (int)e;
}
#endif

// gml_spectconv_bwd_mix_relu with the rows of wmix in TWO arrays: rows [0, nmix_a) from wmix_a, rows [nmix_a, nmix_a + nmix_b) from
// wmix_b -- an ML3Layer's fc11.weight and fc12.weight as they are stored (the single-array form needs their concatenation: a
// launch per layer and step at the reference's batch size)
extern "C" int gml_spectconv_bwd_mix_relu2(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                                           const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                                           float* dx, int64_t lddx, float* dval, float* dw, const float* dz, const float* wmix_a,
                                           int32_t nmix_a, const float* wmix_b, int32_t nmix_b, int32_t relu_cols, int64_t num_rows,
                                           int32_t S, int32_t Fin, int32_t Fout, int32_t max_group_edges, int32_t max_group_window,
                                           uint32_t flags, void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (relu_cols < 0 || relu_cols > Fin || nmix_a < 0 || nmix_b < 0) return GML_E_BADARG;
    const int nmix = nmix_a + nmix_b;
    if (!dz || !wmix_a || (nmix_b > 0 && !wmix_b) || !dx || !gml_spectconv_bwd_mix_supported(S, Fin, Fout, nmix, flags)) return GML_E_UNSUPPORTED;
    if (relu_cols > 0 && (bwd4_env() || (flags & GML_DMA_RING))) return GML_E_UNSUPPORTED;
    return spectconv_bwd_impl(rowptr, col, ginfo, val, x, ldx, g, ldg, w, dx, lddx, dval, dw, num_rows, S, Fin, Fout,
                              max_group_edges, max_group_window, flags, ws, ws_bytes, stream, dz, wmix_a, nmix, relu_cols, wmix_b, nmix_a);
}

// ---------------------------------------------------------------------------------------------
// gml_spectconv_bwd_mix_relu2 with the WHOLE output stage of the ML3Layer inside (libs/spect_conv.py:209-212, backward): instead of
// gml_ml3_split_bwd_ex (pre-masked form) + this kernel, ONE launch.  g [num_rows, ldg >= 32] is the pre-masked gradient at the layer
// output: columns [0, 30) at the conv output, columns 30, 31 at the two Hadamard units.  Shape class: S = 8, 17 .. 32 input
// features (multiple of 4), Fout = 30, F2 = 2 (Zinc12k.py:338-341).  Per workgroup one partial [dw11 | dw12 | db11 | db12 | dcb]
// lands in hws (gml_spectconv_bwd_had_parts rows of 4 Fin + 4 + Fout floats); with every one of dcb .. db12 NULL they stay there
// (gml_fold_many), otherwise they are folded in fixed order into the given ones.
// ---------------------------------------------------------------------------------------------
int gml_split_fold_launch(const float* ws, int64_t nparts, int npart, float* dw11, int n11, float* dw12, int n12, float* db11, int nb11,
                          float* db12, int nb12, float* dcb, int ncb, hipStream_t st);

static bool had_env() { static const bool v = [] { const char* e = getenv("GML_BWD_HAD"); return !(e && e[0] == '0'); }(); return v; }

extern "C" int gml_spectconv_bwd_had_parts(int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout, int32_t F2, int32_t want_dx,
                                           int32_t max_group_edges, int32_t max_group_window, uint32_t flags) {
    /* (without dx -- the model's first layer -- the input width need not be a multiple of 4: only the x rows must be float4-readable) */
    if (!had_env() || num_rows <= 0 || F2 != 2 || Fout != 30 || !gml_spectconv_bwd_mix_supported(S, want_dx ? Fin : (Fin + 3) / 4 * 4, Fout, 4, flags)) return 0;
    if (bwd4_env() || bwd5_env() || (flags & (GML_DMA_RING | GML_ACCUM | GML_DVAL_ACCUM))) return 0;
    const BwdPlan pl = plan_bwd(num_rows, S, Fin, Fout, max_group_edges, max_group_window, flags);
    if (!pl.ok || pl.layout != 3 || pl.nw != 8 || pl.lds + GML_BWD3_HAD_LDS(128, 8) > 160 * 1024) return 0;
    return pl.grid;
}

extern "C" int gml_spectconv_bwd_had(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                                     const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                                     float* dx, int64_t lddx, float* dval, float* dw, const float* w11, const float* b11,
                                     const float* w12, const float* b12, int32_t relu_cols, float* dcb, float* dw11, float* db11,
                                     float* dw12, float* db12, int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout, int32_t F2,
                                     int32_t max_group_edges, int32_t max_group_window, uint32_t flags, void* ws, size_t ws_bytes,
                                     void* hws, size_t hws_bytes, gml_stream_t stream) {
    if (relu_cols < 0 || relu_cols > Fin || ldg < Fout + F2 || (!dx && relu_cols != 0)) return GML_E_BADARG;
    const int parts = gml_spectconv_bwd_had_parts(num_rows, S, Fin, Fout, F2, dx != nullptr, max_group_edges, max_group_window, flags);
    if (parts == 0 || !w11 || !w12 || !dw) return GML_E_UNSUPPORTED;
    if (!dx && !((ldx % 4 == 0) && (((uintptr_t)x & 15) == 0) && (Fin + 3) / 4 * 4 <= ldx)) return GML_E_UNSUPPORTED;   /* float4-readable x rows */
    const int npart = 4 * Fin + 4 + Fout;
    if (!hws || hws_bytes < sizeof(float) * (size_t)parts * npart) return GML_E_WORKSPACE;
    /* (dz = the kernel's own: any non-NULL, 16-byte aligned pointer selects the DZ road of the dispatch; it is never read) */
    const int rc = spectconv_bwd_impl(rowptr, col, ginfo, val, x, ldx, g, ldg, w, dx, lddx, dval, dw, num_rows, S, Fin, Fout,
                                      max_group_edges, max_group_window, flags, ws, ws_bytes, stream, (const float*)hws, w11, 4, relu_cols, w12, 2,
                                      b11, b12, (float*)hws);
    if (rc != GML_OK) return rc;
    if (!dcb && !dw11 && !db11 && !dw12 && !db12) return GML_OK;
    return gml_split_fold_launch((const float*)hws, parts, npart, dw11, 2 * Fin, dw12, 2 * Fin, db11, 2, db12, 2, dcb, Fout, (hipStream_t)stream);
}
