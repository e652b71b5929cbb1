// Backward of the ML3Layer output stage, one pass over the rows (libs/spect_conv.py:209-212 and their autograd):
//
//   out = cat[ relu(conv(x)) , tanh(fc11 x) * tanh(fc12 x) ]
//
//   G[:, :nout1]   = gy[:, :nout1] * (out[:, :nout1] > 0)      (gradient at the conv output; cols [nout1, ldg) = 0)
//   dcb            = column sums of G                           (conv bias gradient)
//   dz1, dz2       = gy[:, nout1:] * tanh' terms                (recomputed from x)
//   dx             = dz1 w11 + dz2 w12                          (WRITTEN; the conv backward then accumulates into it)
//   dw11, dw12, db11, db12 = dz^T [x | 1]
//
// Pre-masked form (y = G = NULL): gy[:, :nout1] already IS G -- the consumer layer's conv backward applied this layer's relu
// mask where it produced the gradient (gml_spectconv_bwd_mix_relu) -- so the saved output is not read and no G is written;
// the pass keeps the bias sums and the Hadamard branch.
//
// Replaces three passes (relu mask, Hadamard-branch backward with a strided read-modify-write of dx, bias
// column sums) by one whose global accesses are all coalesced: SB_ROWS-row tiles, the x / dx tile goes through LDS
// (row-per-lane compute in between), the weight gradients are contracted on the matrix cores as in
// gml_k_node_mix, every sum is formed in a fixed order (per-workgroup partials + fold).
#include "gml_common.h"
#include <stdlib.h>

#ifndef SB_ROWS
#define SB_ROWS 128      /* rows per tile = threads per workgroup: small groups, several per CU, overlap phases */
#endif
#define SB_WAVES (SB_ROWS / 64)
#ifndef SB_WGS_PER_CU
#define SB_WGS_PER_CU 4
#endif
#define SB_MAXCB 3        /* 2*F2 <= 48 */

struct GmlSplitBwdParams {
    const float* gy; int64_t ldgy;
    const int32_t* gyseg;                       // optional: gy has one row per SEGMENT (graph) and row r reads gy[gyseg[r]] -- the
                                                // gradient of a global add / mean pool that directly follows the layer, never expanded
    const float* y; int64_t ldy;
    const float* x; int64_t ldx;
    const float* w11; const float* b11; const float* w12; const float* b12;
    float* G; int64_t ldg;
    float* dx; int64_t lddx;
    float* dz;                                  // optional (2 F2 <= 4): dz[row][0..3] = (dz11 | dz12) instead of dx (gml_spectconv_bwd_mix)
    float* part;
    int64_t nrows;
    int Fin, nout1, F2, ntiles, CP, npart;      // CP: power of two >= ldg (<= 256); npart: floats per partial
    int nopipe;                                 // GML_SPLIT_PIPE=0: the unpipelined loop (A/B)
};

// FINP = 0: no Hadamard branch (plain SpectConv + relu): mask and bias sums only
// VEC / VX = 4: the leading dimensions of gy, y, G / of x, dx are multiples of 4 floats and the bases 16-byte
// aligned (float4 accesses); 1 otherwise
// DZO: the dz hand-over form (dz out, no dx; 2 F2 <= 4: one block of Hadamard columns) -- the ZINC path; only this form is pipelined
// MM (round 6; wide Hadamard branches: counting.py's 16, sr25.py's 16, mutag.py's 24 units): the two projections of the stage --
//   z = X Wc^T (pre-activations) and dx = dz Wc -- run on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products -- a
//   bf16x3 z is not good enough) like the weight-gradient contraction, each wave on its own 64 rows.  The row-per-lane
//   form reads every weight through a wave-wide LDS broadcast (2 x 2 x F2 x FINP / 4 16-byte reads per row: 1,152 at mutag's 24 + 24 /
//   48 inputs) and was bound by exactly that: 0.33 ms per launch at 0.19 of the HBM roof.  Layout changes for it: the w12 units start
//   at a 16-aligned column F2P (a unit and its partner land in the same lane of the D tiles), the Hadamard gradients g sit in the
//   dz tile until dz replaces them (no gi tile: two workgroups per CU still fit), leading dimensions = 4 or 20 mod 32 (conflict-free
//   or 2-way for both operand patterns).
template <int FINP, int VEC, int VX, bool DZO = false, bool MM = false>
__global__ __launch_bounds__((MM ? 2 : 1) * SB_ROWS, (FINP <= 32 || MM ? 2 : 1)) void gml_k_ml3_split_bwd(const GmlSplitBwdParams p) {
    // MM: 256 threads share a 128-row tile -- every per-wave phase works on 32 rows instead of 64, so two workgroups per CU are two
    // waves per SIMD on the same LDS (one wave per SIMD left every load and LDS latency of the phases exposed)
    constexpr int NT = (MM ? 2 : 1) * SB_ROWS, NWV = NT / 64, RPW = SB_ROWS / NWV;
    static_assert(!MM || (!DZO && FINP >= 16 && VEC == 4 && VX == 4), "the matrix-core form is compiled for float4-addressable rows");
    constexpr int MAXCB = DZO ? 1 : (MM ? 4 : SB_MAXCB);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LDX = MM ? FINP + (FINP % 32 == 16 ? 4 : 20) : FINP + 1;   // odd: row-per-lane accesses are conflict free
    constexpr int LDW = MM ? LDX : FINP;                       // rows of the weight image
    constexpr int NFB = FINP / 16;
    const int F2 = p.F2, Fin = p.Fin, nout1 = p.nout1;
    const bool premasked = p.y == nullptr;                     // (uniform) gy's conv columns are G already
    const int F2P = MM ? (F2 + 15) / 16 * 16 : F2;             // first w12 unit's column
    const int C2 = 2 * F2, C2L = MM ? 2 * F2P : C2, C2P = (C2L + 15) / 16 * 16, LDZ = C2P + (MM ? 4 : 1), ncb = C2P / 16, LDI = F2 | 1;
    float* xs = lds;                                           // [SB_ROWS][LDX]   x tile, later the dx tile
    float* wc = xs + SB_ROWS * LDX;                            // [C2L][LDW]   w11 rows then w12 rows, zero padded
    float* bc = wc + C2L * LDW;                                // [C2L]
    float* gz = bc + ((C2L + 3) / 4 * 4);                      // [SB_ROWS][LDZ]   dz1 | dz2
    float* gi = gz + (FINP ? SB_ROWS * LDZ : 0);               // [SB_ROWS][LDI]   gy[:, nout1:]   (MM: inside gz)
    float* red = FINP ? gi + (MM ? 0 : SB_ROWS * LDI) : lds;   // [SB_ROWS]    bias partial fold
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    // pass A: lane <-> VEC consecutive columns, RPS rows per sweep
    const int CPV = p.CP / VEC, RPS = NT / CPV, NIT = SB_ROWS / RPS;
    const int fa = (tid & (CPV - 1)) * VEC, ra = tid / CPV;
    const int Cy = nout1 + F2;                                 // live columns of gy
    const int fa_y = min(fa, (Cy - 1) / VEC * VEC), fa_o = min(fa, (nout1 - 1) / VEC * VEC);   // clamped: loads stay inside
    const int ldgy = (int)p.ldgy, ldy = (int)p.ldy, ldg = (int)p.ldg, ldx = (int)p.ldx, lddx = (int)p.lddx;
    if constexpr (FINP > 0) {
        for (int i = tid; i < C2L * FINP; i += NT) {
            const int c = i / FINP, f = i % FINP;
            const int u = c < F2P ? c : c - F2P;               // unit inside its half (MM: u >= F2 is padding)
            const float v = (f < Fin && u < F2) ? ((c < F2P) ? p.w11[u * Fin + f] : p.w12[u * Fin + f]) : 0.f;
            wc[c * LDW + f] = v;
        }
        for (int i = tid; i < C2L; i += NT) {
            const int u = i < F2P ? i : i - F2P;
            bc[i] = u < F2 ? ((i < F2P) ? (p.b11 ? p.b11[u] : 0.f) : (p.b12 ? p.b12[u] : 0.f)) : 0.f;
        }
    }
    float bacc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) bacc[k] = 0.f;
    f32x4 acc[MAXCB][NFB + 1];
#pragma unroll
    for (int a = 0; a < MAXCB; ++a)
#pragma unroll
        for (int b = 0; b <= NFB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Pipelined form (the ZINC / counting shape class: Fin <= 32 with the Hadamard branch, float4-addressable rows of <= 32
    // columns, no segment map): the NEXT tile's x, gy and y rows are requested before pass B of the current one and committed at
    // the top of the next trip -- with 2 waves per workgroup and 4 workgroups per CU the two exposed load latencies per tile
    // were most of the kernel's time (the arithmetic of a tile is ~1.5 us of a measured 8.7 us per trip).
    constexpr bool PIPE_OK = DZO && FINP > 0 && FINP <= 32 && VEC == 4 && VX == 4;
    constexpr int PNX = PIPE_OK ? FINP / 4 : 1;                // x loads per thread and tile (= NIX below)
    constexpr bool pipe = PIPE_OK;                             // (the dispatcher sends only CP = 32 launches to the DZO instantiation)
    f32x4 px[PNX], pg[PIPE_OK ? 8 : 1], py[PIPE_OK ? 8 : 1];
    auto issue = [&](int tn) {                                 // (clamped, unconditional: 8 + 16 loads in flight)
        if constexpr (PIPE_OK) {
            const int64_t r0n = (int64_t)tn * SB_ROWS;
            const int nrn = (int)min((int64_t)SB_ROWS, p.nrows - r0n);
            constexpr int FV = FINP / 4, RPX = SB_ROWS / FV;
            const float* xb = p.x + r0n * p.ldx;
            const int fc = min((tid % FV) * 4, (Fin - 1) / 4 * 4), rb = tid / FV;
#pragma unroll
            for (int j = 0; j < PNX; ++j) px[j] = *reinterpret_cast<const f32x4*>(xb + __umul24(min(rb + j * RPX, nrn - 1), ldx) + fc);
            const float* gyb = p.gyseg ? p.gy : p.gy + r0n * p.ldgy;
            const float* yb = p.y + r0n * p.ldy;
            int sg[8];                                         // segment map: the row's graph first, then that graph's gradient row
#pragma unroll
            for (int j = 0; j < 8; ++j) sg[j] = p.gyseg ? p.gyseg[r0n + min(ra + j * RPS, nrn - 1)] : min(ra + j * RPS, nrn - 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rr = min(ra + j * RPS, nrn - 1);
                pg[j] = *reinterpret_cast<const f32x4*>(gyb + __umul24(sg[j], ldgy) + fa_y);
                if (!premasked) py[j] = *reinterpret_cast<const f32x4*>(yb + __umul24(rr, ldy) + fa_o);
            }
        }
    };
    if constexpr (pipe) { if ((int)blockIdx.x < p.ntiles) issue(blockIdx.x); }

    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const int64_t r0 = (int64_t)t * SB_ROWS;
        const int nr = (int)min((int64_t)SB_ROWS, p.nrows - r0);
        __syncthreads();                                       // previous tile's readers of xs / gz / gi are done
        // tile-relative 32-bit element offsets from uniform bases (row < 2^8, ld < 2^24: 24-bit multiply-add)
        const float* gyb = p.gyseg ? p.gy : p.gy + r0 * p.ldgy;
        const float* yb = p.y + r0 * p.ldy;
        float* Gb = p.G + r0 * p.ldg;
        if constexpr (pipe) {
            {                                                  // commit what was requested one trip ago
                constexpr int FV = FINP / 4, RPX = SB_ROWS / FV;
                const int f = (tid % FV) * 4, rb = tid / FV;
#pragma unroll
                for (int j = 0; j < PNX; ++j) {
                    const int rr = rb + j * RPX;
#pragma unroll
                    for (int k = 0; k < 4; ++k) xs[rr * LDX + f + k] = (rr < nr && f + k < Fin) ? px[j][k] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int rr = ra + j * RPS;
                    const bool rv = rr < nr;
                    f32x4 gm;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        gm[k] = (rv && fa + k < nout1 && (premasked || py[j][k] > 0.f)) ? pg[j][k] : 0.f;
                        bacc[k] += gm[k];
                        if (fa + k >= nout1 && fa + k < Cy) gi[rr * LDI + (fa + k - nout1)] = rv ? pg[j][k] : 0.f;
                    }
                    if (rv && fa < ldg && !premasked) *reinterpret_cast<f32x4*>(Gb + __umul24(rr, ldg) + fa) = gm;
                }
            }
        } else {
        if constexpr (FINP > 0) {                              // x tile: lane <-> VX features, 8 / 16 loads in flight
            constexpr int FV = FINP / VX, RPX = NT / FV, NIX = (SB_ROWS + RPX - 1) / RPX;
            constexpr int XCH = 8;
            const float* xb = p.x + r0 * p.ldx;
            const int f = (tid % FV) * VX, rb = tid / FV, fc = min(f, (Fin - 1) / VX * VX);
#pragma unroll
            for (int j0 = 0; j0 < NIX; j0 += XCH) {
                float v[XCH][VX];
#pragma unroll
                for (int j = 0; j < XCH; ++j) {
                    const int off = __umul24(min(rb + (j0 + j) * RPX, nr - 1), ldx) + fc;
                    if constexpr (VX == 4) {
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(xb + off);
                        v[j][0] = t4.x; v[j][1] = t4.y; v[j][2] = t4.z; v[j][3] = t4.w;
                    } else {
                        v[j][0] = xb[off];
                    }
                }
#pragma unroll
                for (int j = 0; j < XCH; ++j) {
                    const int rr = rb + (j0 + j) * RPX;
                    if (j0 + j < NIX && rr < SB_ROWS) {
#pragma unroll
                        for (int k = 0; k < VX; ++k) xs[rr * LDX + f + k] = (rr < nr && f + k < Fin) ? v[j][k] : 0.f;
                    }
                }
            }
        }
        // ---- pass A: relu mask, G tile, bias sums, Hadamard-branch gradient columns -> LDS
        constexpr int SB_CHUNK = VEC == 4 ? 8 : 16;            // NIT = CP / VEC is a multiple of it
        for (int j0 = 0; j0 < NIT; j0 += SB_CHUNK) {
            float vg[SB_CHUNK][VEC], vo[SB_CHUNK][VEC];
#pragma unroll
            for (int j = 0; j < SB_CHUNK; ++j) {               // clamped addresses: unconditional, batched loads
                {
                    const int rr = min(ra + (j0 + j) * RPS, nr - 1);
                    const int og = (p.gyseg ? p.gyseg[r0 + rr] * ldgy : __umul24(rr, ldgy)) + fa_y, oy = __umul24(rr, ldy) + fa_o;
                    if constexpr (VEC == 4) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(gyb + og);
                        const f32x4 b4 = premasked ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(yb + oy);
                        vg[j][0] = a4.x; vg[j][1] = a4.y; vg[j][2] = a4.z; vg[j][3] = a4.w;
                        vo[j][0] = b4.x; vo[j][1] = b4.y; vo[j][2] = b4.z; vo[j][3] = b4.w;
                    } else {
                        vg[j][0] = gyb[og];
                        vo[j][0] = premasked ? 1.f : yb[oy];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < SB_CHUNK; ++j) {
                if (j0 + j < NIT) {                             // (MM at narrow rows: fewer sweeps than a chunk)
                    const int rr = ra + (j0 + j) * RPS;
                    const bool rv = rr < nr;
                    float gm[VEC];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        gm[k] = (rv && fa + k < nout1 && vo[j][k] > 0.f) ? vg[j][k] : 0.f;
                        bacc[k] += gm[k];
                        if constexpr (FINP > 0)
                            if (fa + k >= nout1 && fa + k < Cy) (MM ? gz[rr * LDZ + (fa + k - nout1)] : gi[rr * LDI + (fa + k - nout1)]) = rv ? vg[j][k] : 0.f;
                    }
                    if (rv && fa < ldg && !premasked) {
                        const int off = __umul24(rr, ldg) + fa;
                        if constexpr (VEC == 4) *reinterpret_cast<f32x4*>(Gb + off) = f32x4{gm[0], gm[1], gm[2], gm[3]};
                        else Gb[off] = gm[0];
                    }
                }
            }
        }
        }   // (!pipe)
        if constexpr (MM) {                                    // padding units: g = 0, so that 0 * garbage cannot appear (read by the row's wave below)
            if (tid < SB_ROWS)
                for (int c = F2; c < F2P; ++c) gz[tid * LDZ + c] = 0.f;
        }
        if constexpr (FINP > 0) {
            __syncthreads();
            if constexpr (pipe) { if (t + (int)gridDim.x < p.ntiles) issue(t + gridDim.x); }   // the next tile's rows travel during pass B
            // ---- pass B: one row per lane
            float xr[MM ? 1 : FINP], dxr[(DZO || MM) ? 1 : FINP];
            if constexpr (MM) {
                // ---- pass B on the matrix cores: Z^T = Wc X^T for the wave's own 64 rows, 16 at a time.  D[i = unit][j = row]: lane
                //      (r16, kq) ends with units 16 ub + 4 kq + reg of row 16 rb + r16 -- and, from the second product, with their
                //      partners F2P + (the same): tanh, its derivative and dz in place, g read from / dz written to the lane's own 16
                //      bytes of the dz tile
                const int rbase = wave * RPW, nub = F2P / 16;
#pragma unroll 1
                for (int rbk = 0; rbk < RPW / 16; ++rbk) {
                    const int row = rbase + rbk * 16 + r16;
                    for (int ub = 0; ub < nub; ++ub) {
                        f32x4 z1 = f32x4{0.f, 0.f, 0.f, 0.f}, z2 = f32x4{0.f, 0.f, 0.f, 0.f};
                        const float* a1p = wc + (ub * 16 + r16) * LDW + kq;
                        const float* a2p = wc + (F2P + ub * 16 + r16) * LDW + kq;
                        const float* bp = xs + row * LDX + kq;
#pragma unroll
                        for (int ks = 0; ks < FINP / 4; ++ks) {
                            const float b = bp[4 * ks];
                            z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1p[4 * ks], b, z1, 0, 0, 0);
                            z2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2p[4 * ks], b, z2, 0, 0, 0);
                        }
                        float* gq = gz + row * LDZ + ub * 16 + 4 * kq;
                        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gq);
                        const f32x4 ba = *reinterpret_cast<const f32x4*>(bc + ub * 16 + 4 * kq);
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(bc + F2P + ub * 16 + 4 * kq);
                        f32x4 d1, d2;
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            float ta, tb, da, db;
                            gml_tanh_d(z1[reg] + ba[reg], ta, da);
                            gml_tanh_d(z2[reg] + bb[reg], tb, db);
                            d1[reg] = g4[reg] * tb * da;
                            d2[reg] = g4[reg] * ta * db;
                        }
                        *reinterpret_cast<f32x4*>(gq) = d1;
                        *reinterpret_cast<f32x4*>(gq + F2P) = d2;
                    }
                }
            } else {
#pragma unroll
            for (int f = 0; f < FINP; ++f) { xr[f] = xs[tid * LDX + f]; if constexpr (!DZO) dxr[f] = 0.f; }
            for (int c = C2; c < C2P; ++c) gz[tid * LDZ + c] = 0.f;
            for (int o = 0; o < F2; ++o) {
                float a = bc[o], b = bc[F2 + o];
                const float* wa = wc + o * FINP;
                const float* wb = wc + (F2 + o) * FINP;
#pragma unroll
                for (int f4 = 0; f4 < FINP / 4; ++f4) {
                    const f32x4 va = *reinterpret_cast<const f32x4*>(wa + 4 * f4);
                    const f32x4 vb = *reinterpret_cast<const f32x4*>(wb + 4 * f4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a = fmaf(xr[4 * f4 + i], va[i], a);
                        b = fmaf(xr[4 * f4 + i], vb[i], b);
                    }
                }
                float ta, tb, da, db;                          // (tanh and its derivative without the cancellation of 1 - t * t)
                gml_tanh_d(a, ta, da);
                gml_tanh_d(b, tb, db);
                const float g = gi[tid * LDI + o];
                const float g1 = g * tb * da, g2 = g * ta * db;
                gz[tid * LDZ + o] = g1;
                gz[tid * LDZ + F2 + o] = g2;
                if constexpr (!DZO) {
#pragma unroll
                    for (int f4 = 0; f4 < FINP / 4; ++f4) {
                        const f32x4 va = *reinterpret_cast<const f32x4*>(wa + 4 * f4);
                        const f32x4 vb = *reinterpret_cast<const f32x4*>(wb + 4 * f4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) dxr[4 * f4 + i] = fmaf(g1, va[i], fmaf(g2, vb[i], dxr[4 * f4 + i]));
                    }
                }
            }
            }   // (!MM)
            // weight / bias gradients of this wave's 64 rows: D[c][f] += sum_rows dz[row][c] * [x | 1][row][f]
            // (the wave reads only its own rows of gz and xs, written by its own lanes above)
            const int rb = wave * RPW;
#pragma unroll
            for (int t16 = 0; t16 < RPW / 4; ++t16) {
                const int rr = rb + 4 * t16 + kq;
#pragma unroll
                for (int cb = 0; cb < MAXCB; ++cb) {
                    if (cb < ncb) {
                        const float a = gz[rr * LDZ + cb * 16 + r16];
#pragma unroll
                        for (int fb = 0; fb < NFB; ++fb)
                            acc[cb][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xs[rr * LDX + fb * 16 + r16], acc[cb][fb], 0, 0, 0);
                        acc[cb][NFB] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, (r16 == 0) ? 1.f : 0.f, acc[cb][NFB], 0, 0, 0);
                    }
                }
            }
            if (p.dz != nullptr) {                             // hand dz over instead of dx: 16 bytes per row, one lane per row
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < C2) v[c] = gz[tid * LDZ + c];
                if (tid < nr) *reinterpret_cast<f32x4*>(p.dz + (r0 + tid) * 4) = v;
            }
            if (!DZO && p.dx != nullptr) {                     // dx tile through LDS: row-per-lane in, coalesced out
                if constexpr (MM) {
                    // dX^T = Wc^T dZ^T for the wave's own rows (x of those rows is no longer needed: the contraction above was its
                    // last reader): D[i = f][j = row], A[i = f][k = unit] = wc[unit][f], B[k = unit][j = row] = dz[row][unit]
                    const int rbase = wave * RPW;
#pragma unroll 1
                    for (int rbk = 0; rbk < RPW / 16; ++rbk) {
                        const int row = rbase + rbk * 16 + r16;
                        f32x4 dxa[NFB];
#pragma unroll
                        for (int fb = 0; fb < NFB; ++fb) dxa[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
                        const float* bp = gz + row * LDZ + kq;
                        const float* ap = wc + kq * LDW + r16;
                        for (int ks = 0; ks < C2P / 4; ++ks) {
                            const float b = bp[4 * ks];
#pragma unroll
                            for (int fb = 0; fb < NFB; ++fb)
                                dxa[fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * ks * LDW + fb * 16], b, dxa[fb], 0, 0, 0);
                        }
#pragma unroll
                        for (int fb = 0; fb < NFB; ++fb) *reinterpret_cast<f32x4*>(xs + row * LDX + fb * 16 + 4 * kq) = dxa[fb];
                    }
                } else {
#pragma unroll
                for (int f = 0; f < FINP; ++f) xs[tid * LDX + f] = dxr[(DZO || MM) ? 0 : f];
                }
                __syncthreads();
                constexpr int FV = FINP / VX, RPX = NT / FV, NIX = (SB_ROWS + RPX - 1) / RPX;
                float* dxb = p.dx + r0 * p.lddx;
                const int f = (tid % FV) * VX, rbx = tid / FV;
#pragma unroll 8
                for (int j = 0; j < NIX; ++j) {
                    const int rr = rbx + j * RPX;
                    if (rr < nr && f < Fin) {
                        const int off = __umul24(rr, lddx) + f;
                        if constexpr (VX == 4)
                            *reinterpret_cast<f32x4*>(dxb + off) = f32x4{xs[rr * LDX + f], xs[rr * LDX + f + 1],
                                                                         xs[rr * LDX + f + 2], xs[rr * LDX + f + 3]};
                        else dxb[off] = xs[rr * LDX + f];
                    }
                }
            }
        }
    }
    // ---- one partial per workgroup:  [dw11 | dw12 | db11 | db12 | dcb], fixed-order sums
    __syncthreads();
    float* P = p.part + (int64_t)blockIdx.x * p.npart;
    if constexpr (FINP > 0) {
        // D layout: lane (col j = r16, rows i = 4*kq + reg): i = c index, j = f index;  4 waves folded through xs
        constexpr int NA = MAXCB * (NFB + 1) * 4;
        float* wred = xs;                                      // [SB_WAVES][NA][64]
#pragma unroll
        for (int cb = 0; cb < MAXCB; ++cb)
#pragma unroll
            for (int fb = 0; fb <= NFB; ++fb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) wred[(wave * NA + (cb * (NFB + 1) + fb) * 4 + reg) * 64 + lane] = acc[cb][fb][reg];
        __syncthreads();
        for (int it = tid; it < NA * 64; it += NT) {
            const int a = it >> 6, ln = it & 63;
            float v = wred[a * 64 + ln];
#pragma unroll
            for (int w = 1; w < NWV; ++w) v += wred[(w * NA + a) * 64 + ln];
            const int reg = a & 3, fb = (a >> 2) % (NFB + 1), cb = (a >> 2) / (NFB + 1);
            int c = cb * 16 + 4 * (ln >> 4) + reg;
            const int j = ln & 15;
            if constexpr (MM) {                                // padded unit columns -> rows of [dw11; dw12]
                const int u = c < F2P ? c : c - F2P;
                if (c >= C2L || u >= F2) continue;
                c = c < F2P ? u : F2 + u;
            }
            if (c >= C2) continue;
            if (fb < NFB) {
                const int f = fb * 16 + j;
                if (f < Fin) P[c * Fin + f] = v;               // rows c < F2: dw11, then dw12
            } else if (j == 0) {
                P[C2 * Fin + c] = v;                           // db11 | db12
            }
        }
        __syncthreads();
    }
    // thread (ra, fa) holds the sums of columns fa .. fa+VEC-1 over its rows: red[ra][column]
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[ra * p.CP + fa + k] = bacc[k];
    __syncthreads();
    if (tid < nout1) {                                         // nout1 <= CP <= SB_ROWS
        float s = 0.f;
        for (int k = 0; k < RPS; ++k) s += red[k * p.CP + tid];
        P[C2 * Fin + C2 + tid] = s;
    }
}

// fold [nparts][n] partials in fixed order and split into the five destinations
__global__ __launch_bounds__(256) void gml_k_split_fold(const float* __restrict__ partial, int64_t nparts, int n,
                                                       float* __restrict__ d0, int n0, float* __restrict__ d1, int n1,
                                                       float* __restrict__ d2, int n2, float* __restrict__ d3, int n3,
                                                       float* __restrict__ d4, int n4) {
    __shared__ float red[16][17];
    const int jl = threadIdx.x & 15, wl = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + jl;
    float a = 0.f;
    if (j < n) a = gml_fold_column(partial, nparts, n, j, wl);
    red[wl][jl] = a;
    __syncthreads();
    if (wl != 0 || j >= n) return;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][jl];
    int q = j;
    if (q < n0) { if (d0) d0[q] = t; return; }
    q -= n0;
    if (q < n1) { if (d1) d1[q] = t; return; }
    q -= n1;
    if (q < n2) { if (d2) d2[q] = t; return; }
    q -= n2;
    if (q < n3) { if (d3) d3[q] = t; return; }
    q -= n3;
    if (q < n4 && d4) d4[q] = t;
}

// the fold alone, for partial rows another kernel produced in this layout (gml_spectconv_bwd_had)
int gml_split_fold_launch(const float* ws, int64_t nparts, int npart, float* dw11, int n11, float* dw12, int n12, float* db11, int nb11,
                          float* db12, int nb12, float* dcb, int ncb, hipStream_t st) {
    hipLaunchKernelGGL(gml_k_split_fold, dim3((unsigned)gml_cdiv(npart, 16)), dim3(256), 0, st, ws, nparts, npart, dw11, n11, dw12, n12,
                       db11, nb11, db12, nb12, dcb, ncb);
    return gml_launch_status();
}

static int sb_finp(int Fin, int F2) {
    if (F2 == 0) return 0;
    return Fin <= 16 ? 16 : (Fin <= 32 ? 32 : (Fin <= 48 ? 48 : (Fin <= 64 ? 64 : -1)));
}
static int sb_cp(int64_t ldg) { return ldg <= 32 ? 32 : (ldg <= 64 ? 64 : (ldg <= 128 ? 128 : ((ldg <= 256 && SB_ROWS >= 256) ? 256 : 0))); }
static int sb_grid(int64_t num_rows) {
    const int64_t nt = gml_cdiv(num_rows, SB_ROWS);
    static const int wgs = [] { const char* e = getenv("GML_SPLIT_WGS"); const int n = e ? atoi(e) : SB_WGS_PER_CU; return n < 1 ? 1 : n; }();
    return (int)(nt < GML_NUM_CU * wgs ? nt : GML_NUM_CU * wgs);
}
static int sb_npart(int Fin, int nout1, int F2) { return 2 * F2 * Fin + 2 * F2 + nout1; }
static size_t sb_lds(int FINP, int F2, bool mm = false) {
    const int C2 = 2 * F2, C2P = (C2 + 15) / 16 * 16;
    size_t fl = (mm ? 2 : 1) * SB_ROWS * 4;                                 // red (VEC <= 4 columns per thread)
    if (mm) {                                                               // (the layout of the MM instantiation: see the kernel)
        const int F2P = (F2 + 15) / 16 * 16, C2L = 2 * F2P, LDX = FINP + (FINP % 32 == 16 ? 4 : 20);
        fl += (size_t)SB_ROWS * LDX + (size_t)C2L * LDX + C2L + (size_t)SB_ROWS * (C2L + 4);
        const size_t fold = (size_t)2 * SB_WAVES * 4 * (FINP / 16 + 1) * 4 * 64;
        return sizeof(float) * (fl > fold ? fl : fold);
    }
    if (FINP > 0) fl += (size_t)SB_ROWS * (FINP + 1) + (size_t)C2 * FINP + (C2 + 3) / 4 * 4 + (size_t)SB_ROWS * (C2P + 1) +
                        (size_t)SB_ROWS * (F2 | 1);
    const size_t fold = FINP > 0 ? (size_t)SB_WAVES * SB_MAXCB * (FINP / 16 + 1) * 4 * 64 : 0;   // wred aliases xs.. : must fit
    return sizeof(float) * (fl > fold ? fl : fold);
}

extern "C" size_t gml_ml3_split_bwd_workspace_bytes(int64_t num_rows, int32_t Fin, int32_t nout1, int32_t F2) {
    if (num_rows <= 0 || nout1 <= 0 || F2 < 0 || (F2 > 0 && (Fin <= 0 || sb_finp(Fin, F2) < 0 || 2 * F2 > 16 * SB_MAXCB)))
        return 0;                                                            /* 0 = not supported */
    return sizeof(float) * (size_t)sb_grid(num_rows) * sb_npart(F2 ? Fin : 0, nout1, F2);
}

static int split_bwd_impl(const float* gy, int64_t ldgy, const int32_t* gy_seg, const float* y, int64_t ldy, const float* x, int64_t ldx,
                          const float* w11, const float* b11, const float* w12, const float* b12, float* G,
                          int64_t ldg, float* dx, int64_t lddx, float* dz, float* dcb, float* dw11, float* db11, float* dw12,
                          float* db12, int64_t num_rows, int32_t Fin, int32_t nout1, int32_t F2, void* ws,
                          size_t ws_bytes, gml_stream_t stream) {
    if (dz != nullptr && (dx != nullptr || F2 < 1 || 2 * F2 > 4 || (((uintptr_t)dz) & 15) != 0)) return GML_E_BADARG;
    if (num_rows < 0 || nout1 <= 0 || F2 < 0 || ldgy < nout1 + F2 || (y && ldy < nout1) || (G && ldg < nout1)) return GML_E_BADARG;
    const bool nofold = !dcb && !dw11 && !db11 && !dw12 && !db12;   /* no destination at all: the partials stay in ws (gml_fold_many) */
    if (F2 > 0 && (Fin <= 0 || ldx < Fin || (dx && lddx < Fin) || !w11 || !w12 || (!nofold && (!dw11 || !dw12)))) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (F2 == 0) Fin = 0;
    if (num_rows == 0) {
        if (dcb) gml_zero_async(dcb, sizeof(float) * nout1, st);
        if (F2 > 0 && !nofold) {
            gml_zero_async(dw11, sizeof(float) * F2 * Fin, st);
            gml_zero_async(dw12, sizeof(float) * F2 * Fin, st);
            if (db11) gml_zero_async(db11, sizeof(float) * F2, st);
            if (db12) gml_zero_async(db12, sizeof(float) * F2, st);
        }
        return gml_launch_status();
    }
    if (!gy || (y == nullptr) != (G == nullptr) || (F2 > 0 && !x)) return GML_E_BADARG;   /* y = G = NULL: pre-masked gy */
    if (!y && gy_seg) return GML_E_BADARG;
    if (!y) { ldy = ldgy; ldg = ldgy; }
    const int FINP = sb_finp(Fin, F2), CP = sb_cp(ldg > nout1 + F2 ? ldg : nout1 + F2);
    if (FINP < 0 || CP == 0 || 2 * F2 > 16 * SB_MAXCB) return GML_E_UNSUPPORTED;
    const size_t need = gml_ml3_split_bwd_workspace_bytes(num_rows, Fin, nout1, F2);
    if (!ws || ws_bytes < need) return GML_E_WORKSPACE;
    GmlSplitBwdParams p;
    p.gy = gy; p.ldgy = ldgy; p.gyseg = gy_seg; p.y = y; p.ldy = ldy; p.x = x; p.ldx = ldx;
    p.w11 = w11; p.b11 = b11; p.w12 = w12; p.b12 = b12;
    p.G = G; p.ldg = ldg; p.dx = dx; p.lddx = lddx; p.dz = dz; p.part = (float*)ws; p.nrows = num_rows;
    p.Fin = Fin; p.nout1 = nout1; p.F2 = F2; p.ntiles = (int)gml_cdiv(num_rows, SB_ROWS); p.CP = CP;
    p.npart = sb_npart(Fin, nout1, F2);
    { static const int np = [] { const char* e = getenv("GML_SPLIT_PIPE"); return (e && e[0] == '0') ? 1 : 0; }(); p.nopipe = np; }
    const int grid = sb_grid(num_rows);
    /* wide Hadamard branches on float4-addressable rows: the projections on the matrix cores (GML_SPLIT_MM=0: the row-per-lane form) */
    static const bool mm_env = [] { const char* e = getenv("GML_SPLIT_MM"); return !(e && e[0] == '0'); }();
    const bool vec44 = (((ldgy | ldy | ldg) & 3) == 0 && (((uintptr_t)gy | (uintptr_t)y | (uintptr_t)G) & 15) == 0) &&
                       (F2 == 0 || (((ldx | (dx ? lddx : 0)) & 3) == 0 && (((uintptr_t)x | (uintptr_t)dx) & 15) == 0));
    const bool mm = mm_env && F2 >= 16 && F2 <= 32 && FINP >= 32 && vec44 && dz == nullptr && sb_lds(FINP, F2, true) <= 160 * 1024;
    const size_t lds = sb_lds(FINP, F2, mm);
    const int va = (((ldgy | ldy | ldg) & 3) == 0 && (((uintptr_t)gy | (uintptr_t)y | (uintptr_t)G) & 15) == 0) ? 4 : 1;
    const int vx = (F2 == 0 || (((ldx | (dx ? lddx : 0)) & 3) == 0 && (((uintptr_t)x | (uintptr_t)dx) & 15) == 0)) ? 4 : 1;
#define SB_GO(FP, VA, VXX)                                                                                       \
    if (FINP == FP && va == VA && vx == VXX) {                                                                   \
        GML_ALLOW_BIG_LDS(arc, (&gml_k_ml3_split_bwd<FP, VA, VXX>), 160 * 1024) \
        if (arc != hipSuccess) return (int)arc;                                                                  \
        hipLaunchKernelGGL((gml_k_ml3_split_bwd<FP, VA, VXX>), dim3(grid), dim3(SB_ROWS), lds, st, p);           \
    }
#define SB_GO4(FP) SB_GO(FP, 1, 1) SB_GO(FP, 1, 4) SB_GO(FP, 4, 1) SB_GO(FP, 4, 4)
    /* the dz hand-over form on float4-addressable rows (Zinc12k.py's layers): its own, pipelined instantiation */
    const bool dzo = dz != nullptr && dx == nullptr && 2 * F2 <= 4 && va == 4 && vx == 4 && (FINP == 16 || FINP == 32) && CP == 32 && !p.nopipe;
#define SB_GO_DZ(FP)                                                                                             \
    if (dzo && FINP == FP) {                                                                                     \
        GML_ALLOW_BIG_LDS(arc, (&gml_k_ml3_split_bwd<FP, 4, 4, true>), 160 * 1024)                               \
        if (arc != hipSuccess) return (int)arc;                                                                  \
        hipLaunchKernelGGL((gml_k_ml3_split_bwd<FP, 4, 4, true>), dim3(grid), dim3(SB_ROWS), lds, st, p);        \
    }
    SB_GO_DZ(16) SB_GO_DZ(32)
#define SB_GO_MM(FP)                                                                                             \
    if (mm && FINP == FP) {                                                                                      \
        GML_ALLOW_BIG_LDS(arc, (&gml_k_ml3_split_bwd<FP, 4, 4, false, true>), 160 * 1024)                        \
        if (arc != hipSuccess) return (int)arc;                                                                  \
        hipLaunchKernelGGL((gml_k_ml3_split_bwd<FP, 4, 4, false, true>), dim3(grid), dim3(2 * SB_ROWS), lds, st, p); \
    }
    SB_GO_MM(16) SB_GO_MM(32) SB_GO_MM(48) SB_GO_MM(64)
    if (!dzo && !mm) {
    SB_GO(0, 1, 4) SB_GO(0, 4, 4) SB_GO4(16) SB_GO4(32) SB_GO4(48) SB_GO4(64)
    }
    int rc = gml_launch_status();
    if (rc != GML_OK) return rc;
    if (nofold) return GML_OK;                               /* nothing to fold into: the partials stay in ws (gml_fold_many) */
    hipLaunchKernelGGL(gml_k_split_fold, dim3((unsigned)gml_cdiv(p.npart, 16)), dim3(256), 0, st, (const float*)ws,
                       (int64_t)grid, p.npart, dw11, F2 * Fin, dw12, F2 * Fin, db11, F2, db12, F2, dcb, nout1);
    return gml_launch_status();
}

extern "C" int gml_ml3_split_bwd(const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                                 const float* w11, const float* b11, const float* w12, const float* b12, float* G,
                                 int64_t ldg, float* dx, int64_t lddx, float* dcb, float* dw11, float* db11, float* dw12,
                                 float* db12, int64_t num_rows, int32_t Fin, int32_t nout1, int32_t F2, void* ws,
                                 size_t ws_bytes, gml_stream_t stream) {
    return split_bwd_impl(gy, ldgy, nullptr, y, ldy, x, ldx, w11, b11, w12, b12, G, ldg, dx, lddx, nullptr, dcb, dw11, db11, dw12, db12,
                          num_rows, Fin, nout1, F2, ws, ws_bytes, stream);
}

// the general form.  dz != NULL (then dx == NULL; 2 F2 <= 4): the Hadamard branch's share of dx is handed over as
// dz [num_rows, 4] = (dz11 | dz12) (columns beyond 2 F2 zero) for gml_spectconv_bwd_mix -- dx = dz [w11; w12] is formed
// inside the conv backward and never stored.  gy_seg != NULL: gy has one row per segment (graph) and row r reads
// gy[gy_seg[r]]: the gradient of the global add pool that follows the last layer (Zinc12k.py:343) is never expanded to
// [num_rows, C] (for a mean pool the caller divides the pooled gradient by the segment sizes first).
extern "C" int gml_ml3_split_bwd_ex(const float* gy, int64_t ldgy, const int32_t* gy_seg, const float* y, int64_t ldy,
                                    const float* x, int64_t ldx, const float* w11, const float* b11, const float* w12,
                                    const float* b12, float* G, int64_t ldg, float* dx, int64_t lddx, float* dz, float* dcb,
                                    float* dw11, float* db11, float* dw12, float* db12, int64_t num_rows, int32_t Fin,
                                    int32_t nout1, int32_t F2, void* ws, size_t ws_bytes, gml_stream_t stream) {
    return split_bwd_impl(gy, ldgy, gy_seg, y, ldy, x, ldx, w11, b11, w12, b12, G, ldg, dx, lddx, dz, dcb, dw11, db11, dw12, db12,
                          num_rows, Fin, nout1, F2, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// out[i][j] = sum_r A[r][i] * B[r][j]   (a, b <= 64; n rows in the millions): the weight gradient of a small
// dense layer applied to every row (readout head fc1 / fc2 of Zinc12k.py:343-345).  A library GEMM maps this
// M = a, N = b, K = n problem onto ONE workgroup; here the rows are split over the chip, each wave contracts
// its 64 rows on the matrix cores (v_mfma_f32_16x16x4_f32, both tiles staged coalesced through LDS) and the
// per-workgroup partials are folded in fixed order.
// ---------------------------------------------------------------------------------------------
#define XTY_ROWS 256
static __host__ __device__ inline int xty_ld(int a) { return a <= 16 ? 16 : (a <= 48 ? 48 : 80); }   // = 16 or 48 mod 64

__global__ __launch_bounds__(256) void gml_k_xty(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                 int64_t ldb, float* __restrict__ part, int64_t n, int a, int b,
                                                 int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LDA = xty_ld(a), LDB = xty_ld(b);
    float* As = lds;                         // [256][LDA]
    float* Bs = As + XTY_ROWS * LDA;         // [256][LDB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int nab = (a + 15) / 16, nbb = (b + 15) / 16;
    f32x4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t r0 = (int64_t)t * XTY_ROWS;
        const int nr = (int)min((int64_t)XTY_ROWS, n - r0);
        __syncthreads();
        // coalesced tile loads, eight (clamped, unconditional) loads in flight per thread and operand
        for (int i0 = tid; i0 < XTY_ROWS * LDA; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u, r = i / LDA, c = i - r * LDA;
                v[u] = A[(r0 + min(r, nr - 1)) * lda + min(c, a - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u, r = i / LDA, c = i - r * LDA;
                if (i < XTY_ROWS * LDA) As[i] = (r < nr && c < a) ? v[u] : 0.f;
            }
        }
        for (int i0 = tid; i0 < XTY_ROWS * LDB; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u, r = i / LDB, c = i - r * LDB;
                v[u] = B[(r0 + min(r, nr - 1)) * ldb + min(c, b - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u, r = i / LDB, c = i - r * LDB;
                if (i < XTY_ROWS * LDB) Bs[i] = (r < nr && c < b) ? v[u] : 0.f;
            }
        }
        __syncthreads();
        const int rb = wave * 64;
#pragma unroll 4
        for (int t16 = 0; t16 < 16; ++t16) {
            const int rr = rb + 4 * t16 + kq;
            float av[4], bv[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) av[x] = (x < nab) ? As[rr * LDA + x * 16 + i16] : 0.f;
#pragma unroll
            for (int y = 0; y < 4; ++y) bv[y] = (y < nbb) ? Bs[rr * LDB + y * 16 + i16] : 0.f;
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
                    if (x < nab && y < nbb) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
    }
    // D[x][y]: lane (col = i16, rows 4*kq + reg): row = A column, col = B column; fold the 4 waves through LDS
    __syncthreads();
    float* red = lds;                        // [4 waves][16 blocks][4 regs][64 lanes] = 64 KB max; only live blocks used
    const int nblk = nab * nbb;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
            if (x < nab && y < nbb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) red[((wave * nblk + x * nbb + y) * 4 + reg) * 64 + lane] = acc[x][y][reg];
    __syncthreads();
    float* P = part + (int64_t)blockIdx.x * a * b;
    for (int it = tid; it < nblk * 4 * 64; it += 256) {
        const int ln = it & 63, reg = (it >> 6) & 3, blk = it >> 8;
        const float v = ((red[((0 * nblk + blk) * 4 + reg) * 64 + ln] + red[((1 * nblk + blk) * 4 + reg) * 64 + ln]) +
                         red[((2 * nblk + blk) * 4 + reg) * 64 + ln]) + red[((3 * nblk + blk) * 4 + reg) * 64 + ln];
        const int x = blk / nbb, y = blk - x * nbb;
        const int i = x * 16 + 4 * (ln >> 4) + reg, j = y * 16 + (ln & 15);
        if (i < a && j < b) P[i * b + j] = v;
    }
}

static int xty_grid(int64_t n) {
    const int64_t nt = gml_cdiv(n, XTY_ROWS);
    return (int)(nt < GML_NUM_CU * 2 ? nt : GML_NUM_CU * 2);
}

extern "C" size_t gml_xty_workspace_bytes(int64_t n, int32_t a, int32_t b) {
    if (n <= 0 || a <= 0 || b <= 0 || a > 64 || b > 64) return 0;
    return sizeof(float) * (size_t)xty_grid(n) * a * b;
}

extern "C" int gml_xty(const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t n, int32_t a,
                       int32_t b, void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (n < 0 || a <= 0 || b <= 0 || lda < a || ldb < b || !out) return GML_E_BADARG;
    if (a > 64 || b > 64) return GML_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        gml_zero_async(out, sizeof(float) * a * b, st);
        return gml_launch_status();
    }
    if (!A || !B) return GML_E_BADARG;
    if (!ws || ws_bytes < gml_xty_workspace_bytes(n, a, b)) return GML_E_WORKSPACE;
    const int grid = xty_grid(n);
    size_t lds = sizeof(float) * (size_t)XTY_ROWS * (xty_ld(a) + xty_ld(b));
    const size_t fold = sizeof(float) * (size_t)4 * ((a + 15) / 16) * ((b + 15) / 16) * 4 * 64;
    if (fold > lds) lds = fold;
    GML_ALLOW_BIG_LDS(arc, (&gml_k_xty), 160 * 1024)
    if (arc != hipSuccess) return (int)arc;
    hipLaunchKernelGGL(gml_k_xty, dim3(grid), dim3(256), lds, st, A, lda, B, ldb, (float*)ws, n, a, b,
                       (int)gml_cdiv(n, XTY_ROWS));
    int rc = gml_launch_status();
    if (rc != GML_OK) return rc;
    hipLaunchKernelGGL(gml_k_split_fold, dim3((unsigned)gml_cdiv(a * b, 16)), dim3(256), 0, st, (const float*)ws,
                       (int64_t)grid, a * b, out, a * b, (float*)nullptr, 0, (float*)nullptr, 0, (float*)nullptr, 0,
                       (float*)nullptr, 0);
    return gml_launch_status();
}
