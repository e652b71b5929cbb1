// Dispatch of the fused forward kernel families + the unfused SpMM / SDDMM kernels.
#include "gml_spectconv_impl.h"
#include "gml_spectconv_fwd2_impl.h"
#include "gml_spectconv_fwd3_impl.h"
#include "gml_spectconv_fwd4_impl.h"
#include "gml_spmm3_impl.h"

#define GML_DECL_FWD2(S, B) template <> int gml_launch_fwd2<S, B>(const GmlFwdParams&, dim3, hipStream_t, bool, bool);
GML_DECL_FWD2(8, 2) GML_DECL_FWD2(8, 1) GML_DECL_FWD2(4, 2) GML_DECL_FWD2(4, 1)
GML_DECL_FWD2(8, 0) GML_DECL_FWD2(4, 0)            /* NOB = 0: the stand-alone SpMM instantiation */
GML_DECL_FWD2(12, 2) GML_DECL_FWD2(12, 1) GML_DECL_FWD2(12, 0)  /* counting.py's 12 supports */

#define GML_DECL_FWD3(S, B) template <> int gml_launch_fwd3<S, B>(const GmlFwdParams&, dim3, hipStream_t, bool);
GML_DECL_FWD3(8, 2) GML_DECL_FWD3(8, 1) GML_DECL_FWD3(4, 2) GML_DECL_FWD3(4, 1)

#if GML_F4DBG & (8 | 64)
static unsigned long long* f4dbg_buf() {
    static unsigned long long* b = [] { unsigned long long* q = nullptr; (void)hipMalloc(&q, 64); (void)hipMemset(q, 0, 64); return q; }();
    return b;
}
extern "C" int gml_debug_f4_counts(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpy(out, f4dbg_buf(), 64, hipMemcpyDeviceToHost);
    if (e == hipSuccess && reset) e = hipMemset(f4dbg_buf(), 0, 64);
    return (int)e;
}
#endif
#define GML_DECL_FWD4(S, FB, B) template <> int gml_launch_fwd4<S, FB, B>(const GmlFwdParams&, dim3, hipStream_t);
GML_DECL_FWD4(4, 0, 2) GML_DECL_FWD4(4, 1, 2) GML_DECL_FWD4(6, 0, 2) GML_DECL_FWD4(6, 1, 2) GML_DECL_FWD4(8, 0, 2)

// GML_FWD_DMA=0: the register-staged 8-wave kernel (fwd2) instead of the LDS-DMA ring (fwd3), for A/B runs
static bool fwd3_env() { static const bool v = [] { const char* e = getenv("GML_FWD_DMA"); return !(e && e[0] == '0'); }(); return v; }

// shapes only the chunked ring kernel (fwd4) serves on 128-row records: 6 supports (sr25.py) and / or 33 .. 48 input features
// (the hidden layers of sr25.py:252-262 and mutag.py:272-288); x must be float4-addressable there
static bool fwd4_only_shape(int S, int Fin, int Fout, uint32_t flags) {
    static const bool off = [] { const char* e = getenv("GML_FWD4"); return e && e[0] == '0'; }();   // A/B: the r03 roads
    if (off || (flags & GML_F32_MFMA) || Fout > 32) return false;
    return (S == 6 && Fin <= 48) || (S == 4 && Fin > 32 && Fin <= 48);     /* (8 supports x 48 features: no script has it; its instantiation spilled) */
}
// GML_FWD4=1: every shape of the ring kernels on the chunked one (A/B against fwd3)
static bool fwd4_all_env() { static const bool v = [] { const char* e = getenv("GML_FWD4"); return e && e[0] == '1'; }(); return v; }

static bool fwd2_shape(int S, int Fin, int Fout, uint32_t flags) {
#ifdef GML_NO_FWD2
    return false;
#endif
    // S % 4 == 0: the register-staged value rows are float4 (other S keep the 64-row kernel, which stages any S)
    return ((flags & GML_F32_MFMA) == 0 && (S == 4 || S == 8 || S == 12) && Fin <= 32 && Fout <= 32) || fwd4_only_shape(S, Fin, Fout, flags);
}

// GML_FWD_NW=4: the 8-wave kernel family in its 4-wave / 64-row geometry (two workgroups per CU)
static int fwd2_nw_env() { static const int v = [] { const char* e = getenv("GML_FWD_NW"); return e ? atoi(e) : 8; }(); return v == 4 ? 4 : 8; }
extern "C" int32_t gml_spectconv_fwd_group_rows(int32_t S, int32_t Fin, int32_t Fout, uint32_t flags) {
    if (!fwd2_shape(S, Fin, Fout, flags)) return 64;
    return fwd2_nw_env() == 4 ? GML_GROUPS64_RANKED : 128;
}

// edges of one 128-row group the ring kernel of this shape keeps in LDS at once (one work item); 0: the shape is on no ring kernel.
// Larger groups: fwd3 shapes gather them from global memory, or -- GML_FWD_CHUNKED -- run on the chunked ring kernel; the shapes only
// the chunked kernel serves walk them in edge chunks (functional.FWD_CHUNKS: on by default since the fix of DESIGN s4.1c).
extern "C" int32_t gml_spectconv_fwd_stage_edges(int32_t S, int32_t Fin, int32_t Fout, uint32_t flags) {
    if (!fwd2_shape(S, Fin, Fout, flags)) return 0;
    if (fwd4_only_shape(S, Fin, Fout, flags)) {                /* one work item of the chunked ring kernel (its gathering form: the smaller one) */
        const int fb = Fin > 32 ? 1 : 0;
        if (S == 6) return (fb ? GmlFwd4Cfg<6, 1, true>::ECAP : GmlFwd4Cfg<6, 0, true>::ECAP) - 3;
        return GmlFwd4Cfg<4, 1, true>::ECAP - 3;
    }
    if (!fwd3_env()) return 0;
    return S == 8 ? GmlFwd3Cfg<8>::ECAP - 3 : (S == 4 ? GmlFwd3Cfg<4>::ECAP - 3 : 0);
}

// widest column window (int 3 of a 128-row group record) the chunked ring kernel serves for this shape; 0 = no bound (the shape's
// kernel has its own road for wider groups).  Batches with a wider group must stay on the 64-row family (group_rows 64): the chunked
// kernel has no global-gather road and marks the rows of such a group NaN.
extern "C" int32_t gml_spectconv_fwd_stage_window(int32_t S, int32_t Fin, int32_t Fout, uint32_t flags) {
    if (!fwd2_shape(S, Fin, Fout, flags)) return 0;
    if (!(fwd4_only_shape(S, Fin, Fout, flags) || (flags & GML_FWD_CHUNKED) || fwd4_all_env())) return 0;
    return Fin > 32 ? GmlFwd4Cfg<4, 1, false>::XCAP - 15 : GmlFwd4Cfg<4, 0, false>::XCAP - 7;
}

// ---- families defined in gml_fwd_fam_*.hip ---------------------------------------------------
#define GML_DECL_FAM(SC, FPL) \
    template <> int gml_launch_fwd_family<SC, FPL>(const GmlFwdParams&, int, bool, bool, dim3, size_t, hipStream_t);
GML_DECL_FAM(1, 8) GML_DECL_FAM(2, 8) GML_DECL_FAM(3, 8) GML_DECL_FAM(4, 8) GML_DECL_FAM(6, 8) GML_DECL_FAM(8, 8)
GML_DECL_FAM(1, 4) GML_DECL_FAM(2, 4) GML_DECL_FAM(3, 4) GML_DECL_FAM(4, 4) GML_DECL_FAM(6, 4) GML_DECL_FAM(8, 4)
GML_DECL_FAM(12, 4) GML_DECL_FAM(16, 4)

static int launch_family(int SC, int FPL, const GmlFwdParams& p, int NB, bool xvec, bool bf, dim3 grid, size_t lds,
                         hipStream_t st) {
#define GML_FAM(SCV, FPLV) \
    if (SC == SCV && FPL == FPLV) return gml_launch_fwd_family<SCV, FPLV>(p, NB, xvec, bf, grid, lds, st);
    GML_FAM(1, 8) GML_FAM(2, 8) GML_FAM(3, 8) GML_FAM(4, 8) GML_FAM(6, 8) GML_FAM(8, 8)
    GML_FAM(1, 4) GML_FAM(2, 4) GML_FAM(3, 4) GML_FAM(4, 4) GML_FAM(6, 4) GML_FAM(8, 4)
    GML_FAM(12, 4) GML_FAM(16, 4)
    return GML_E_UNSUPPORTED;
}

static const int kSC8[] = {8, 6, 4, 3, 2, 1};
static const int kSC4[] = {16, 12, 8, 6, 4, 3, 2, 1};

extern "C" int gml_node_mix_fwd(const float* x, int64_t ldx, const float* w11, const float* b11, const float* w12,
                                const float* b12, float* out, int64_t ldo, int64_t num_rows, int32_t Fin,
                                int32_t F2, gml_stream_t stream);

#ifdef GML_FWD2_TIMING
static unsigned long long* fwd2_prof_buf() {
    static unsigned long long* b = [] { unsigned long long* q = nullptr; (void)hipMalloc(&q, 256); (void)hipMemset(q, 0, 256); return q; }();
    return b;
}
extern "C" int gml_debug_fwd2_prof(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpy(out, fwd2_prof_buf(), 256, hipMemcpyDeviceToHost);   /* [0,16): fwd2, [16,32): fwd3 */
    if (e == hipSuccess && reset) e = hipMemset(fwd2_prof_buf(), 0, 256);
    return (int)e;
}
#endif

// conv (+ optionally the Hadamard branch of the same rows) on the 8-wave kernel; 128-row group records
static int launch_fwd2(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const int32_t* epos, const float* val, const float* x,
                       int64_t ldx, const float* w, int64_t w_ss, int64_t w_si, int64_t w_so, const float* bias,
                       const float* w11, const float* b11, const float* w12, const float* b12, float* out, int64_t ldo,
                       int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout, int32_t F2, uint32_t flags, hipStream_t st) {
    const bool xv = (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0);
    GmlFwdParams p = {};
    p.rowptr = rowptr; p.col = col; p.ginfo = ginfo; p.epos = epos; p.val = val; p.x = x; p.ldx = ldx;
    p.w = w; p.w_ss = w_ss; p.w_si = w_si; p.w_so = w_so; p.bias = bias; p.out = out; p.ldo = ldo;
    p.nrows = num_rows; p.S = S; p.Fin = Fin; p.Fout = Fout; p.flags = flags; p.s0 = 0; p.npass = 1; p.nchunks = 1;
    p.val_vec = 1;
    p.w11 = w11; p.b11 = b11; p.w12 = w12; p.b12 = b12; p.F2 = F2; p.mix_col = Fout;
#ifdef GML_FWD2_TIMING
    p.prof = fwd2_prof_buf();
#endif
#if GML_F4DBG & (8 | 64)
    p.prof = f4dbg_buf();
#endif
#if GML_F4DBG & 256
    { const char* e = getenv("GML_F4_HOUT"); p.hout = e ? (float*)(uintptr_t)strtoull(e, nullptr, 0) : nullptr; }
#endif
    p.nw = (flags & GML_GROUPS64R) ? 4 : 8;
    if (p.nw == 4 && !xv) return GML_E_UNSUPPORTED;
    const int wgs = p.nw == 4 ? 2 * GML_NUM_CU : GML_NUM_CU;         // one 512-thread or two 256-thread workgroups per CU
    p.ngroups = (int)gml_cdiv(num_rows, 16 * p.nw);
    int grid = p.ngroups < wgs ? p.ngroups : wgs;
    p.groups_per_wg = (int)gml_cdiv(p.ngroups, grid);
    grid = (int)gml_cdiv(p.ngroups, p.groups_per_wg);
    const int nob = Fout > 16 ? 2 : 1;
    const bool mix = F2 > 0;
    // chunked ring kernel (fwd4): the shapes only it serves, groups with more edges than fwd3 stages (GML_FWD_CHUNKED: the caller
    // knows the largest group), or everything (GML_FWD4=1)
    const bool only4 = fwd4_only_shape(S, Fin, Fout, flags);
    if (only4 || ((flags & GML_FWD_CHUNKED) || fwd4_all_env())) {
        const bool ok4 = p.nw == 8 && xv && !mix && !(flags & GML_ACCUM) && (num_rows + 16) * ldx * 4 < (int64_t)INT32_MAX &&
                         (S == 4 || S == 6 || S == 8) && Fin <= 48 && !(S == 8 && Fin > 32);
        if (ok4) {
            const int fb = Fin > 32 ? 1 : 0;
#define GML_FWD4_GO(SV, FBV) if (S == SV && fb == FBV) return gml_launch_fwd4<SV, FBV, 2>(p, dim3(grid), st);
            GML_FWD4_GO(4, 0) GML_FWD4_GO(4, 1) GML_FWD4_GO(6, 0) GML_FWD4_GO(6, 1) GML_FWD4_GO(8, 0)
        }
        if (only4) return GML_E_UNSUPPORTED;             /* (unaligned x rows, accumulate mode: the caller takes the 64-row family) */
    }
    // LDS-DMA landing ring: float4-addressable x, no accumulate mode, 32-bit buffer offsets
    // (value rows beyond 4 GB are handled inside the kernel: it reads the edge count itself)
    if (p.nw == 8 && xv && fwd3_env() && (S == 8 || S == 4) && !(flags & GML_ACCUM) &&
        (num_rows + 16) * ldx * 4 < (int64_t)INT32_MAX) {
#define GML_FWD3_GO(SV, B) if (S == SV && nob == B) return gml_launch_fwd3<SV, B>(p, dim3(grid), st, mix);
        GML_FWD3_GO(8, 2) GML_FWD3_GO(8, 1) GML_FWD3_GO(4, 2) GML_FWD3_GO(4, 1)
    }
#define GML_FWD2_GO(SV, B) if (S == SV && nob == B) return gml_launch_fwd2<SV, B>(p, dim3(grid), st, xv, mix);
    GML_FWD2_GO(8, 2) GML_FWD2_GO(8, 1) GML_FWD2_GO(4, 2) GML_FWD2_GO(4, 1) GML_FWD2_GO(12, 2) GML_FWD2_GO(12, 1)
    return GML_E_UNSUPPORTED;
}

// SpectConCatConv / depthwise SpectConv forward on the ring kernel's epilogues (see GmlFwdParams::epl).  epilogue = 1: out has
// (S + self_term) column blocks of Fout, support s writes block s + self_term (+ bias of that block; block 0 of a selfconn layer,
// x W_last, is the caller's GEMM); epilogue = 2: w is ONE [Fin, Fout] matrix (w_ss ignored), ds [S + self_term, Fin] the
// per-feature scales (row 0 = 1 + DSweight[0]; last row = the self term's scale when self_term): out = (sum_s ds_s . H_s +
// ds_self . x) W + bias.  GML_E_UNSUPPORTED outside the kernel's shape class (S in {4, 8}, Fin, Fout <= 32, float4-addressable
// x): the caller then uses the weight-transform mapping onto gml_spectconv_fwd.
extern "C" int gml_spectconv_fwd_epi(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo128, const float* val,
                                     const float* x, int64_t ldx, const float* w, int64_t w_ss, int64_t w_si, int64_t w_so,
                                     const float* bias, float* out, int64_t ldo, int64_t num_rows, int32_t S, int32_t Fin,
                                     int32_t Fout, uint32_t flags, int32_t epilogue, const float* ds, int32_t self_term,
                                     gml_stream_t stream) {
    if (num_rows < 0 || S <= 0 || Fin <= 0 || Fout <= 0 || ldx < Fin) return GML_E_BADARG;
    if (epilogue != 1 && epilogue != 2) return GML_E_BADARG;
    if (epilogue == 2 && ds == nullptr) return GML_E_BADARG;
    if (ldo < (epilogue == 1 ? (int64_t)(S + (self_term ? 1 : 0)) * Fout : Fout)) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!rowptr || !ginfo128 || !x || !w || !out) return GML_E_BADARG;
    const bool xv = (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0);
    if (!fwd2_shape(S, Fin, Fout, flags) || !(S == 8 || S == 4) || !xv || !fwd3_env() || (flags & GML_ACCUM) ||
        (((uintptr_t)val & 15) != 0) || (num_rows + 16) * ldx * 4 >= (int64_t)INT32_MAX)
        return GML_E_UNSUPPORTED;
    GmlFwdParams p = {};
    p.rowptr = rowptr; p.col = col; p.ginfo = ginfo128; p.val = val; p.x = x; p.ldx = ldx;
    p.w = w; p.w_ss = w_ss; p.w_si = w_si; p.w_so = w_so; p.bias = bias; p.out = out; p.ldo = ldo;
    p.nrows = num_rows; p.S = S; p.Fin = Fin; p.Fout = Fout; p.flags = flags; p.npass = 1; p.nchunks = 1; p.val_vec = 1;
    p.nw = 8; p.epl = epilogue; p.ds = ds; p.ds_self = self_term ? 1 : 0; p.cc_off = self_term ? 1 : 0;
    p.ngroups = (int)gml_cdiv(num_rows, 128);
    int grid = p.ngroups < GML_NUM_CU ? p.ngroups : GML_NUM_CU;
    p.groups_per_wg = (int)gml_cdiv(p.ngroups, grid);
    grid = (int)gml_cdiv(p.ngroups, p.groups_per_wg);
    const int nob = Fout > 16 ? 2 : 1;
    hipStream_t st = (hipStream_t)stream;
#define GML_FWD3_EPI(SV, B) if (S == SV && nob == B) return gml_launch_fwd3<SV, B>(p, dim3(grid), st, false);
    GML_FWD3_EPI(8, 2) GML_FWD3_EPI(8, 1) GML_FWD3_EPI(4, 2) GML_FWD3_EPI(4, 1)
    return GML_E_UNSUPPORTED;
}

extern "C" int gml_spectconv_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const int32_t* epos,
                                 const float* val, const float* x, int64_t ldx,
                                 const float* w, int64_t w_ss, int64_t w_si, int64_t w_so,
                                 const float* bias, float* out, int64_t ldo,
                                 int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                                 uint32_t flags, gml_stream_t stream) {
    if (num_rows < 0 || S <= 0 || Fin <= 0 || Fout <= 0 || ldx < Fin || ldo < Fout) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!rowptr || !ginfo || !x || !w || !out) return GML_E_BADARG;   /* col/val may be null when there are no edges */
    if (num_rows > (int64_t)INT32_MAX - 16) return GML_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;

    if (flags & (GML_GROUPS128 | GML_GROUPS64R)) {
        // 128-row / 8-wave kernel (or its 64-row / 4-wave geometry): the caller passes 128-row group records (gml_spectconv_fwd_group_rows said 128)
        if (!fwd2_shape(S, Fin, Fout, flags) || (((uintptr_t)val & (S % 4 == 0 ? 15 : 7)) != 0)) return GML_E_BADARG;
        return launch_fwd2(rowptr, col, ginfo, epos, val, x, ldx, w, w_ss, w_si, w_so, bias, nullptr, nullptr, nullptr, nullptr,
                           out, ldo, num_rows, S, Fin, Fout, 0, flags, st);
    }

    // features per lane per chunk: 8 unless 4 pads the contraction less
    const int pad8 = (Fin + 31) / 32 * 32, pad4 = (Fin + 15) / 16 * 16;
    // The bf16x3 projection needs 8 features per lane, so outside the exact-fp32 mode the wider padding is taken even where
    // 4 per lane would pad less (Fin = 48: sr25 / mutag GNNML3 forward -22 % / -15 %; GML_FWD_FPL4=1 restores the old choice)
    static const bool fpl4 = [] { const char* e = getenv("GML_FWD_FPL4"); return e && e[0] == '1'; }();
    // (narrow inputs, Fin <= 16, keep 4 per lane = exact products: with so few terms per output the split's 2^-17 is not
    //  averaged and mutag GNNML1's batch-normalised gradients left the 1e-4 bar, 1.5e-4)
    const int FPL = (pad4 < pad8 && (fpl4 || (flags & GML_F32_MFMA) || Fin <= 32)) ? 4 : 8;
    const int CH = 4 * FPL;
    const int nchunks = (Fin + CH - 1) / CH;
    const bool xvec = (Fin % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0);
    const bool val_ok = (((uintptr_t)val & 15) == 0);
    if (!val_ok) return GML_E_BADARG;

    // support passes: one launch if S = npass * SC for a compiled SC, else a greedy split with accumulation
    const int* scs = (FPL == 8) ? kSC8 : kSC4;
    const int nscs = (FPL == 8) ? 6 : 8;
    struct Part { int s0, count, sc; } parts[32];
    int nparts = 0;
    for (int i = 0; i < nscs && nparts == 0; ++i)
        if (S % scs[i] == 0 && (scs[i] >= 4 || scs[i] == S)) { parts[0] = {0, S, scs[i]}; nparts = 1; }
    if (nparts == 0) {
        int s0 = 0;
        while (s0 < S) {
            int sc = 1;
            for (int i = 0; i < nscs; ++i) if (scs[i] <= S - s0) { sc = scs[i]; break; }
            if (nparts == 32) return GML_E_UNSUPPORTED;
            parts[nparts++] = {s0, sc, sc};
            s0 += sc;
        }
    }

    // persistent workgroups, each a contiguous range of 64-row groups; 2 resident per CU (LDS bound)
    const int ngroups = (int)gml_cdiv(num_rows, GML_GROUP);
    int grid = ngroups < GML_NUM_CU * 2 ? ngroups : GML_NUM_CU * 2;
    const int groups_per_wg = (int)gml_cdiv(ngroups, grid);
    grid = (int)gml_cdiv(ngroups, groups_per_wg);

    for (int ip = 0; ip < nparts; ++ip) {
        const int SC = parts[ip].sc;
        const int colgrp = (SC * FPL > 32) ? 64 : 128;   // keep one W block <= 64 KiB of LDS
        for (int o0 = 0; o0 < Fout; o0 += colgrp) {
            const int fo = (Fout - o0 < colgrp) ? Fout - o0 : colgrp;
            const int nb16 = (fo + 15) / 16;
            const int NB = nb16 <= 1 ? 1 : (nb16 <= 2 ? 2 : (nb16 <= 4 ? 4 : 8));
            GmlFwdParams p = {};
            p.rowptr = rowptr; p.col = col; p.ginfo = ginfo; p.epos = epos;
            p.val = val; p.x = x; p.ldx = ldx;
            p.w = w + (int64_t)o0 * w_so; p.w_ss = w_ss; p.w_si = w_si; p.w_so = w_so;
            p.out = out + o0; p.ldo = ldo; p.nrows = num_rows; p.S = S; p.Fin = Fin; p.Fout = fo;
            p.s0 = parts[ip].s0; p.npass = parts[ip].count / SC; p.nchunks = nchunks;
            p.ngroups = ngroups; p.groups_per_wg = groups_per_wg;
            const bool last = (ip == nparts - 1), first = (ip == 0);
            p.bias = (last && bias) ? bias + o0 : nullptr;
            p.flags = (first ? (flags & GML_ACCUM) : GML_ACCUM) | (last ? (flags & GML_RELU) : 0u) | (flags & 0xff00u);
            const size_t wblk = (size_t)SC * FPL * NB * 64 * sizeof(float);
            const size_t all = wblk * p.npass * nchunks;
            const int ecap = SC == 6 ? 2 * GML_ECAP : GML_ECAP;                  /* = GmlStage<SC, FPL>::ECAP */
            const size_t stage = (size_t)(76 + ecap + ecap * SC + GML_XCAP * (4 * FPL + 4)) * sizeof(float);
            p.allw = all + stage <= 80 * 1024;           // two workgroups per CU
            const int va = (SC % 4 == 0) ? 4 : ((SC % 2 == 0) ? 2 : 1);
            p.val_vec = (S % va == 0) && (p.s0 % va == 0);
            p.wfloats = (int)((p.allw ? all : wblk) / sizeof(float));
            const size_t lds = (size_t)p.wfloats * sizeof(float) + stage;
            int rc = launch_family(SC, FPL, p, NB, xvec && FPL >= 4, (flags & GML_F32_MFMA) == 0, dim3(grid), lds, st);
            if (rc != GML_OK) return rc;
        }
    }
    return GML_OK;
}

// ML3Layer forward (libs/spect_conv.py:204-212) minus the edge branch: out[:, :nout1] = relu?(conv(x)) and
// out[:, nout1:nout1+F2] = tanh(fc11 x) * tanh(fc12 x); one launch on the 8-wave kernel when it applies (F2 <= 8).
extern "C" int gml_ml3_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const int32_t* epos,
                           const float* val, const float* x, int64_t ldx, const float* w, int64_t w_ss, int64_t w_si, int64_t w_so,
                           const float* bias, const float* w11, const float* b11, const float* w12, const float* b12,
                           float* out, int64_t ldo, int64_t num_rows, int32_t S, int32_t Fin, int32_t nout1,
                           int32_t F2, uint32_t flags, gml_stream_t stream) {
    if (F2 < 0 || ldo < nout1 + F2 || (F2 > 0 && (!w11 || !w12))) return GML_E_BADARG;
#ifndef GML_NO_MIXFUSE
    if (num_rows > 0 && F2 > 0 && F2 <= 8 && (flags & (GML_GROUPS128 | GML_GROUPS64R)) && fwd2_shape(S, Fin, nout1, flags) &&
        !fwd4_only_shape(S, Fin, nout1, flags) && !(flags & GML_FWD_CHUNKED) && !fwd4_all_env() &&
        (((uintptr_t)val & 15) == 0) && !(flags & GML_ACCUM)) {
        if (!rowptr || !ginfo || !x || !w || !out) return GML_E_BADARG;
        return launch_fwd2(rowptr, col, ginfo, epos, val, x, ldx, w, w_ss, w_si, w_so, bias, w11, b11, w12, b12, out, ldo,
                           num_rows, S, Fin, nout1, F2, flags, (hipStream_t)stream);
    }
#endif
    int rc = gml_spectconv_fwd(rowptr, col, ginfo, epos, val, x, ldx, w, w_ss, w_si, w_so, bias, out, ldo, num_rows, S,
                               Fin, nout1, flags, stream);
    if (rc != GML_OK || F2 == 0) return rc;
    return gml_node_mix_fwd(x, ldx, w11, b11, w12, b12, out + nout1, ldo, num_rows, Fin, F2, stream);
}

// =============================================================================================
// Unfused SpMM: H[r, s, :] = sum_k val[pos(k), s] * x[col[k], :]   (materialises H; used for dW)
// GW lanes share one row (lane <-> feature, coalesced X rows), 64/GW rows per wave.
// =============================================================================================
template <int SC, int GW>
__global__ __launch_bounds__(256) void gml_k_spmm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                 const int32_t* __restrict__ epos, const float* __restrict__ val,
                                                 const float* __restrict__ x, int64_t ldx, float* __restrict__ h,
                                                 int64_t nrows, int S, int s0, int Fin) {
    constexpr int RPW = 64 / GW;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t row = wave * RPW + lane / GW;
    const int lr = lane % GW;
    if (row >= nrows) return;
    const int kbeg = rowptr[row], kend = rowptr[row + 1];
    for (int f = lr; f < Fin; f += GW) {
        float acc[SC];
#pragma unroll
        for (int s = 0; s < SC; ++s) acc[s] = 0.f;
        for (int k = kbeg; k < kend; ++k) {
            const int64_t pk = epos ? (int64_t)epos[k] : (int64_t)k;
            const float xv = x[(int64_t)col[k] * ldx + f];
            const float* vr = val + pk * S + s0;
#pragma unroll
            for (int s = 0; s < SC; ++s) acc[s] = fmaf(vr[s], xv, acc[s]);
        }
#pragma unroll
        for (int s = 0; s < SC; ++s) h[(row * S + s0 + s) * Fin + f] = acc[s];
    }
}

template <int SC>
static int launch_spmm(const int32_t* rowptr, const int32_t* col, const int32_t* epos, const float* val,
                       const float* x, int64_t ldx, float* h, int64_t nrows, int S, int s0, int Fin, hipStream_t st) {
    const int GW = Fin <= 16 ? 16 : (Fin <= 32 ? 32 : 64);
    const int64_t waves = gml_cdiv(nrows, 64 / GW);
    const dim3 grid((unsigned)gml_cdiv(waves, 4));
    if (GW == 16) hipLaunchKernelGGL((gml_k_spmm<SC, 16>), grid, dim3(256), 0, st, rowptr, col, epos, val, x, ldx, h, nrows, S, s0, Fin);
    else if (GW == 32) hipLaunchKernelGGL((gml_k_spmm<SC, 32>), grid, dim3(256), 0, st, rowptr, col, epos, val, x, ldx, h, nrows, S, s0, Fin);
    else hipLaunchKernelGGL((gml_k_spmm<SC, 64>), grid, dim3(256), 0, st, rowptr, col, epos, val, x, ldx, h, nrows, S, s0, Fin);
    return gml_launch_status();
}

extern "C" int gml_spmm_fwd_ex(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo128, const int32_t* epos,
                               const float* val, const float* x, int64_t ldx, float* h, int64_t num_rows, int32_t S,
                               int32_t Fin, int32_t max_group_edges, gml_stream_t stream);
extern "C" int gml_spmm_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo128, const int32_t* epos,
                            const float* val, const float* x, int64_t ldx, float* h, int64_t num_rows, int32_t S,
                            int32_t Fin, gml_stream_t stream) {
    return gml_spmm_fwd_ex(rowptr, col, ginfo128, epos, val, x, ldx, h, num_rows, S, Fin, -1, stream);
}

extern "C" int gml_spmm_fwd_ex(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo128, const int32_t* epos,
                               const float* val, const float* x, int64_t ldx, float* h, int64_t num_rows, int32_t S,
                               int32_t Fin, int32_t max_group_edges, gml_stream_t stream) {
    if (num_rows < 0 || S <= 0 || Fin <= 0 || ldx < Fin) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!rowptr || !x || !h) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    // ring kernel (gml_spmm3_impl.h): any S (chunks of 4, 2, 1 supports), any Fin (chunks of 32 features), degrees up to ~16 per
    // row on average staged in LDS.  GML_SPMM3=0: the r02 paths below, for A/B runs.
    static const bool spmm3_on = [] { const char* e = getenv("GML_SPMM3"); return !(e && e[0] == '0'); }();
    // Shapes the register-staged 8-wave kernel (fwd2, NOB = 0) covers keep it while every group fits its 1024-edge staging
    // (ZINC: 0.65 vs 0.61 of the roof at 131,072 graphs); unknown group sizes (max_group_edges < 0): as before r03.
    const bool fwd2_fits = fwd2_shape(S, Fin, 16, 0) && (((uintptr_t)val & 15) == 0) && (max_group_edges < 0 || max_group_edges <= GML_FWD2_ECAP);
    if (spmm3_on && !fwd2_fits && ginfo128 != nullptr && epos == nullptr && (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0) && Fin % 4 == 0 &&
        (((uintptr_t)h & 15) == 0) && (((uintptr_t)val & 3) == 0) && (num_rows + 16) * ldx * 4 < (int64_t)INT32_MAX &&
        (int64_t)128 * S * Fin * 4 < (int64_t)INT32_MAX && (int64_t)S * 4 * 24 <= GmlSpmm3Cfg::VAL_BYTES) {
        GmlSpmm3Params q = {};
        q.rowptr = rowptr; q.col = col; q.ginfo = ginfo128; q.val = val; q.ldx = ldx; q.h = h; q.nrows = num_rows; q.S = S; q.hs = Fin;
        q.ngroups = (int)gml_cdiv(num_rows, 128);
        int grid = q.ngroups < GML_NUM_CU ? q.ngroups : GML_NUM_CU;
        q.groups_per_wg = (int)gml_cdiv(q.ngroups, grid);
        grid = (int)gml_cdiv(q.ngroups, q.groups_per_wg);
        static const bool w48_on = [] { const char* e = getenv("GML_SPMM3_W48"); return !(e && e[0] == '0'); }();
        for (int f0 = 0; f0 < Fin;) {                           // feature chunks: separate launches, the value rows are read again
            const int left = Fin - f0;
            // a remainder of 36 .. 48 features is ONE launch of the 192-byte-row form (sr25 / mutag hidden width 48); else chunks of 32
            const bool w48 = w48_on && left > 32 && left <= 48 && (int64_t)S * 4 * 24 <= GmlSpmm3CfgT<true>::VAL_BYTES;
            q.x = x + f0; q.Fin = w48 ? left : (left < 32 ? left : 32); q.hf0 = f0;
            int rc;
            if (w48) rc = (S % 4 == 0) ? gml_launch_spmm3<4, true>(q, dim3(grid), st)
                          : ((S % 2 == 0) ? gml_launch_spmm3<2, true>(q, dim3(grid), st) : gml_launch_spmm3<1, true>(q, dim3(grid), st));
            else rc = (S % 4 == 0) ? gml_launch_spmm3<4>(q, dim3(grid), st)
                      : ((S % 2 == 0) ? gml_launch_spmm3<2>(q, dim3(grid), st) : gml_launch_spmm3<1>(q, dim3(grid), st));
            if (rc != GML_OK) return rc;
            f0 += q.Fin;
        }
        return GML_OK;
    }
    if (ginfo128 != nullptr && epos == nullptr && fwd2_shape(S, Fin, 16, 0) && (((uintptr_t)val & 15) == 0) &&
        (((uintptr_t)h & 15) == 0)) {
        // the 8-wave kernel's staged, degree-ranked aggregation; H written straight from the accumulators
        const bool xv = (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0);
        GmlFwdParams p = {};
        p.rowptr = rowptr; p.col = col; p.ginfo = ginfo128; p.val = val; p.x = x; p.ldx = ldx;
        p.nrows = num_rows; p.S = S; p.Fin = Fin; p.Fout = 16; p.npass = 1; p.nchunks = 1; p.val_vec = 1; p.hout = h;
        p.ngroups = (int)gml_cdiv(num_rows, GML_FWD2_ROWS);
        int grid = p.ngroups < GML_NUM_CU ? p.ngroups : GML_NUM_CU;
        p.groups_per_wg = (int)gml_cdiv(p.ngroups, grid);
        grid = (int)gml_cdiv(p.ngroups, p.groups_per_wg);
        if (S == 8) return gml_launch_fwd2<8, 0>(p, dim3(grid), st, xv, false);
        if (S == 4) return gml_launch_fwd2<4, 0>(p, dim3(grid), st, xv, false);
        if (S == 12) return gml_launch_fwd2<12, 0>(p, dim3(grid), st, xv, false);
    }
    int s0 = 0;
    while (s0 < S) {
        const int rem = S - s0;
        int rc;
        if (rem >= 8) { rc = launch_spmm<8>(rowptr, col, epos, val, x, ldx, h, num_rows, S, s0, Fin, st); s0 += 8; }
        else if (rem >= 4) { rc = launch_spmm<4>(rowptr, col, epos, val, x, ldx, h, num_rows, S, s0, Fin, st); s0 += 4; }
        else if (rem >= 2) { rc = launch_spmm<2>(rowptr, col, epos, val, x, ldx, h, num_rows, S, s0, Fin, st); s0 += 2; }
        else { rc = launch_spmm<1>(rowptr, col, epos, val, x, ldx, h, num_rows, S, s0, Fin, st); s0 += 1; }
        if (rc != GML_OK) return rc;
    }
    return GML_OK;
}

// =============================================================================================
// SDDMM: dval[pos(k), s] = < x[col[k], :], gw[r, s, :] >   (gradient of message() w.r.t. norm)
// GW lanes share one row; each lane keeps its slice of gw[r] in registers; width-GW shuffle tree.
// =============================================================================================
template <int SC, int GW, int NFC>
__global__ __launch_bounds__(256) void gml_k_sddmm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                  const int32_t* __restrict__ epos, const float* __restrict__ x,
                                                  int64_t ldx, const float* __restrict__ gw, float* __restrict__ dval,
                                                  int64_t nrows, int S, int s0, int Fin) {
    constexpr int RPW = 64 / GW;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t row = wave * RPW + lane / GW;
    const int lr = lane % GW;
    const bool valid = row < nrows;
    const int kbeg = valid ? rowptr[row] : 0, kend = valid ? rowptr[row + 1] : 0;
    float g[SC][NFC];
#pragma unroll
    for (int s = 0; s < SC; ++s)
#pragma unroll
        for (int c = 0; c < NFC; ++c) {
            const int f = lr + c * GW;
            g[s][c] = (valid && f < Fin) ? gw[(row * S + s0 + s) * Fin + f] : 0.f;
        }
    // all lanes of a wave must run the same trip count for the shuffles
    int n = kend - kbeg;
#pragma unroll
    for (int off = 32; off >= GW; off >>= 1) n = max(n, __shfl_xor(n, off));
    for (int i = 0; i < n; ++i) {
        const int k = kbeg + i;
        const bool kv = k < kend;
        float part[SC];
#pragma unroll
        for (int s = 0; s < SC; ++s) part[s] = 0.f;
        if (kv) {
            const float* xr = x + (int64_t)col[k] * ldx;
#pragma unroll
            for (int c = 0; c < NFC; ++c) {
                const int f = lr + c * GW;
                const float xv = (f < Fin) ? xr[f] : 0.f;
#pragma unroll
                for (int s = 0; s < SC; ++s) part[s] = fmaf(xv, g[s][c], part[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < SC; ++s)
#pragma unroll
            for (int off = GW / 2; off >= 1; off >>= 1) part[s] += __shfl_xor(part[s], off);
        if (kv && lr == 0) {
            const int64_t pk = epos ? (int64_t)epos[k] : (int64_t)k;
#pragma unroll
            for (int s = 0; s < SC; ++s) dval[pk * S + s0 + s] = part[s];
        }
    }
}

template <int SC>
static int launch_sddmm(const int32_t* rowptr, const int32_t* col, const int32_t* epos, const float* x, int64_t ldx,
                        const float* gw, float* dval, int64_t nrows, int S, int s0, int Fin, hipStream_t st) {
    const int GW = Fin <= 16 ? 16 : (Fin <= 32 ? 32 : 64);
    const int64_t waves = gml_cdiv(nrows, 64 / GW);
    const dim3 grid((unsigned)gml_cdiv(waves, 4));
#define GML_SDDMM(GWV, NFCV) \
    hipLaunchKernelGGL((gml_k_sddmm<SC, GWV, NFCV>), grid, dim3(256), 0, st, rowptr, col, epos, x, ldx, gw, dval, nrows, S, s0, Fin)
    if (GW == 16) GML_SDDMM(16, 1);
    else if (GW == 32) GML_SDDMM(32, 1);
    else if (Fin <= 64) GML_SDDMM(64, 1);
    else if (Fin <= 128) GML_SDDMM(64, 2);
    else if (Fin <= 256) GML_SDDMM(64, 4);
    else return GML_E_UNSUPPORTED;
    return gml_launch_status();
}

extern "C" int gml_sddmm(const int32_t* rowptr, const int32_t* col, const int32_t* epos, const float* x, int64_t ldx,
                         const float* gw, float* dval, int64_t num_rows, int32_t S, int32_t Fin, gml_stream_t stream) {
    if (num_rows < 0 || S <= 0 || Fin <= 0 || ldx < Fin) return GML_E_BADARG;
    if (num_rows == 0) return GML_OK;
    if (!rowptr || !x || !gw) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int s0 = 0;
    while (s0 < S) {
        const int rem = S - s0;
        int rc;
        if (rem >= 8) { rc = launch_sddmm<8>(rowptr, col, epos, x, ldx, gw, dval, num_rows, S, s0, Fin, st); s0 += 8; }
        else if (rem >= 4) { rc = launch_sddmm<4>(rowptr, col, epos, x, ldx, gw, dval, num_rows, S, s0, Fin, st); s0 += 4; }
        else if (rem >= 2) { rc = launch_sddmm<2>(rowptr, col, epos, x, ldx, gw, dval, num_rows, S, s0, Fin, st); s0 += 2; }
        else { rc = launch_sddmm<1>(rowptr, col, epos, x, ldx, gw, dval, num_rows, S, s0, Fin, st); s0 += 1; }
        if (rc != GML_OK) return rc;
    }
    return GML_OK;
}
