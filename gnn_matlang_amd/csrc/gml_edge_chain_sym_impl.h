// ML3Layer edge branch over the UNIQUE support rows of a batch (round 6).
//
// The supports SpectralDesign produces are samples of symmetric matrices (U f(L) U^T, powers of A: libs/utils.py:546-610), so the rows
// of edge (i, j) and of its mirror (j, i) are -- for 92 % of ZINC-like edges bit for bit -- the same 32 bytes, and the edge branch
//
//   out = relu( W4 . [ relu(W1 e) ; tanh(W2 e) * tanh(W3 e) ] )                         libs/spect_conv.py:205-207
//
// gives both the same output row.  Common-subexpression elimination on the DATA: gml_edge_sym_flags marks, per batch, the edges whose
// mirror row is bitwise identical (flag 2 on the copy with src < dst, 0 on the other; 1 = evaluate alone: self loops, rows that differ
// -- the +-1e-17 noise of structurally-zero entries --, edges without a mirror).  The forward evaluates the unique rows (62 % of the
// edges) and stores each result to the edge's row and to its mirror's; the backward adds the mirror's output gradient to the edge's
// before ONE pass of the chain -- sum_e go_e (x) h_e with h_e = h_mirror is (go_e + go_mirror) (x) h_e.  Nothing is approximated: an
// edge whose mirror differs in one bit is evaluated on its own, asymmetric inputs simply find no pairs.  Both kernels are
// instruction-issue-bound (DESIGN s4.3), so the skipped evaluations are time saved; the bytes are the same (every output row is
// still written, every gradient row still read).  2 <= S = Sout <= 8; the raw supports carry no gradient on this road.
#pragma once
#include "gml_edge_chain6_impl.h"

// cache policy of the forward's output stores: 0 = default (write-back).  The plain forward streams its rows out with nt (2); here the
// mirror's row is a second, scattered 32-byte store into a line a later entry completes -- keeping the lines in the L2 until they are
// full measured 1.15 vs 1.28 ms per step for the four-layer forward (profiles/r06_d_edge_sym_ab.log)
#ifndef GML_SYM_ST_AUX
#define GML_SYM_ST_AUX 0
#endif
// workgroups per CU of the backward (launch bound): 3 = 150 VGPRs, no spills; 4 = 128 VGPRs + 8 spilled: 1.83 vs 1.45 ms per step
#ifndef GML_SYM_BWD_WGS
#define GML_SYM_BWD_WGS 3
#endif

// flag / mirror of every edge of the SOURCE-keyed view (row r = source, col_t = targets ascending inside a row).  2^LPS lanes share a
// source row and deal its edges (the row is known without a search; ZINC-like rows hold ~6 edges: 4 lanes per row), the mirror is
// found by bisection in the target's row
__global__ __launch_bounds__(256) void gml_k_edge_sym_flags(const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ col_t,
                                                           const uint32_t* __restrict__ val, int64_t N, int64_t E, int S, int LPS,
                                                           int32_t* __restrict__ flag, int32_t* __restrict__ mirror) {
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t src = gt >> LPS;
    if (src >= N) return;
    const int lpr = 1 << LPS;
    const int k0 = rowptr_t[src], k1 = rowptr_t[src + 1];
    for (int k = k0 + (int)(gt & (lpr - 1)); k < k1; k += lpr) {
        const int dst = col_t[k];
        int f = 1, m = -1;
        // (a multigraph's repeated edge (i, j) has no unique mirror: such edges -- on either side -- are evaluated alone)
        const bool dup = (k > k0 && col_t[k - 1] == dst) || (k + 1 < k1 && col_t[k + 1] == dst);
        if (src != dst && !dup) {
            const int e1 = rowptr_t[dst + 1];
            int a = rowptr_t[dst], b = e1;                     // find src among the targets of row dst (first occurrence)
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (col_t[mid] < (int)src) a = mid + 1; else b = mid;
            }
            if (a < e1 && col_t[a] == (int)src && !(a + 1 < e1 && col_t[a + 1] == (int)src)) {
                // a unique pair: the copy with src < dst compares the two rows ONCE and writes both records (the other copy reaches this
                // point under exactly the mirrored conditions and leaves its record to it): half the row reads of the pass
                if ((int)src < dst) {
                    bool same = true;
                    const uint32_t* pk = val + (int64_t)k * S;
                    const uint32_t* pa = val + (int64_t)a * S;
                    if ((S & 3) == 0 && (reinterpret_cast<uintptr_t>(val) & 15) == 0) {   // rows of aligned 16-byte groups
                        for (int s4 = 0; s4 < S; s4 += 4) {
                            const u32x4 x = *reinterpret_cast<const u32x4*>(pk + s4), y = *reinterpret_cast<const u32x4*>(pa + s4);
                            same = same && x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w;
                        }
                    } else {
                        for (int s = 0; s < S; ++s) same = same && (pk[s] == pa[s]);
                    }
                    flag[k] = same ? 2 : 1;
                    mirror[k] = same ? a : -1;
                    flag[a] = same ? 0 : 1;
                    mirror[a] = -1;
                }
                continue;
            }
        }
        flag[k] = f;
        mirror[k] = m;
    }
}

// ------------------------------------------------------------------------------------------ forward over the unique rows
// entry u: uid[u] = the edge to evaluate, mir[u] = its mirror (-1: none).  out[l][uid] and out[l][mir] receive the row.
template <int S, int L, bool TA>
__global__ __launch_bounds__(256, 2) void gml_k_edge_chain6_fwd_sym(const float* __restrict__ ea, const int32_t* __restrict__ uid,
                                                                   const int32_t* __restrict__ mir, const GmlChain6Stack<L> a,
                                                                   int64_t E, int64_t U, int64_t ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    GmlChain6W<S> W[L];
#pragma unroll
    for (int l = 0; l < L; ++l) gml_chain6_load_weights<S>(W[l], a.w1[l], a.w2[l], a.w3[l], a.w4[l], c16, g);
    GmlNegI negI;
    gml_chain_make_negI(negI, c16, g);
    const int q0 = 4 * (g & 1);
    const int64_t stride = (int64_t)gridDim.x * 8;
    int64_t t = ((int64_t)blockIdx.x * 4 + wave) * 2;
    constexpr int OOB = 0x7ffffff0;
    // the lane's store row: lane groups 0, 1 write the edge's own row, groups 2, 3 its mirror's; < 0: nothing to store
    auto fetch_idx = [&](int64_t tt, int32_t (&el)[2], int32_t (&st)[2]) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int64_t u = (tt + v) * 16 + c16;
            const int64_t uc = u < U ? u : U - 1;
            const int32_t e = uid[uc], m = mir[uc];
            el[v] = e;
            st[v] = u < U ? (g < 2 ? e : m) : -1;
        }
    };
    auto fetch_rows = [&](const int32_t (&el)[2], float (&r)[2][8]) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const float* p = ea + (int64_t)el[v] * S;
            if constexpr (S % 4 == 0) {
#pragma unroll
                for (int j = 0; j < S / 4; ++j) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(p + 4 * j);
                    r[v][4 * j] = x.x; r[v][4 * j + 1] = x.y; r[v][4 * j + 2] = x.z; r[v][4 * j + 3] = x.w;
                }
            } else if constexpr (S % 2 == 0) {
#pragma unroll
                for (int j = 0; j < S / 2; ++j) {
                    const f32x2 x = *reinterpret_cast<const f32x2*>(p + 2 * j);
                    r[v][2 * j] = x.x; r[v][2 * j + 1] = x.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < S; ++j) r[v][j] = p[j];
            }
#pragma unroll
            for (int j = S; j < 8; ++j) r[v][j] = 0.f;
        }
    };
    int32_t el_c[2], st_c[2], el_n[2], st_n[2], el_nn[2], st_nn[2];
    float ec[2][8], en[2][8];
    fetch_idx(t, el_c, st_c);
    fetch_idx(t + stride, el_n, st_n);
    fetch_rows(el_c, ec);
    for (; t < ntiles; t += stride) {
        bf16x8 BA[2], BB[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) gml_chain6_b1(ec[v], g, BA[v], BB[v]);
        fetch_rows(el_n, en);                                  // next pair's rows (their indices arrived a trip ago)
        fetch_idx(t + 2 * stride, el_nn, st_nn);               // and the indices of the pair after it
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            f32x4 o[2];
#pragma unroll
            for (int v = 0; v < 2; ++v) o[v] = gml_chain6_forward<S, TA, GML_CHAIN6_RES>(W[l], negI, BA[v], BB[v]);
            const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(a.out[l], 0, (int)(uint32_t)(E * S * 4), 0x00020000);
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const f32x4 x = f32x4{gml_relu1(o[v][0]), gml_relu1(o[v][1]), gml_relu1(o[v][2]), gml_relu1(o[v][3])};
                const int off = (st_c[v] >= 0 && q0 < S) ? (st_c[v] * S + q0) * 4 : OOB;
                if constexpr (S % 4 == 0) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), rs_o, off, 0, GML_SYM_ST_AUX);
                } else if constexpr (S == 6) {                   // rows of 24 bytes: columns 0..3 as one 16-byte store (dword-aligned: enough for a
                    typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));                      // buffer store), columns 4, 5 as one 8-byte store
                    if (q0 == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), rs_o, off, 0, GML_SYM_ST_AUX);
                    else __builtin_amdgcn_raw_buffer_store_b64(u32x2_{__float_as_uint(x[0]), __float_as_uint(x[1])}, rs_o, off, 0, GML_SYM_ST_AUX);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x[r]), rs_o, (q0 + r < S && off != OOB) ? off + 4 * r : OOB, 0, GML_SYM_ST_AUX);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
#pragma unroll
            for (int j = 0; j < S; ++j) { asm volatile("" : "+v"(en[v][j])); ec[v][j] = en[v][j]; }
            asm volatile("" : "+v"(el_nn[v]), "+v"(st_nn[v]));
            el_c[v] = el_n[v]; st_c[v] = st_n[v];
            el_n[v] = el_nn[v]; st_n[v] = st_nn[v];
        }
    }
}

template <int S, int L>
int gml_launch_edge_chain6_fwd_sym(const float* ea, const int32_t* uid, const int32_t* mir, const GmlChain6Stack<L>& a, int64_t E,
                                   int64_t U, hipStream_t st) {
    const int64_t ntiles = gml_cdiv(U, 16);
    int64_t grid = gml_cdiv(ntiles, 8);
    if (grid > 4 * GML_NUM_CU) grid = 4 * GML_NUM_CU;
    if (gml_chain6_accurate_tanh())
        hipLaunchKernelGGL((gml_k_edge_chain6_fwd_sym<S, L, true>), dim3((unsigned)grid), dim3(256), 0, st, ea, uid, mir, a, E, U, ntiles);
    else
        hipLaunchKernelGGL((gml_k_edge_chain6_fwd_sym<S, L, false>), dim3((unsigned)grid), dim3(256), 0, st, ea, uid, mir, a, E, U, ntiles);
    return gml_launch_status();
}

// ------------------------------------------------------------------------------------------ backward over the unique rows
// gml_k_edge_chain_bwd<S, GIN = false, PRE = true> (gml_edge_chain_impl.h) with the index indirection: tile entry u evaluates edge
// uid[u] on the output gradient gout[uid[u]] + gout[mir[u]].  Same arithmetic (two-piece chain, recomputed intermediates), same
// partial-sum layout [dw1 | dw2 | dw3 | dw4] per workgroup.
template <int S>
__global__ __launch_bounds__(256, GML_SYM_BWD_WGS) void gml_k_edge_chain_bwd_sym(
    const uint32_t* __restrict__ es, const int32_t* __restrict__ uid, const int32_t* __restrict__ mir, const float* __restrict__ w1,
    const float* __restrict__ w2, const float* __restrict__ w3, const float* __restrict__ w4, const float* __restrict__ gout,
    float* __restrict__ partial, int64_t U, int64_t ntiles) {
    constexpr int H2 = 2 * S, H4 = 4 * S;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 12 * 512];
    float (*red)[20][64] = reinterpret_cast<float (*)[20][64]>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int q0 = 4 * (g & 1);
    unsigned char* trw = smem + wave * (12 * 512);
    const int tr_wo = g * 128 + ((c16 ^ ((g >> 1) << 3)) << 3);
    const int tr_ro = (c16 & 3) * 128 + (((4 * g + (c16 >> 2)) ^ (((c16 & 3) >> 1) << 3)) << 3);
    GmlChainW<S> W;
    gml_chain_load_fwd_weights<S>(W, w1, w2, w3, w4, c16, g);
    bf16x8 a3[2], aE;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + j;
            v[j] = v[4 + j] = (c16 < H2 && q < S) ? w4[q * H4 + blk * H2 + c16] : 0.f;
        }
        a3[blk] = gml_wop(v, g >= 2);
    }
    {
        float ve[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ve[j] = (g >= 2 && c16 == 4 * g + (j & 3)) ? 1.f : 0.f;
        aE = gml_wop(ve, false);
    }
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) acc[b] = zero;

    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t t = (int64_t)blockIdx.x * 4 + wave;
    struct Idx { int32_t e, m; bool ok; };
    auto fetch_idx = [&](int64_t tt) -> Idx {
        const int64_t u = tt * 16 + c16;
        const int64_t uc = u < U ? u : U - 1;
        Idx r;
        r.e = uid[uc]; r.m = mir[uc]; r.ok = u < U;
        return r;
    };
    u32x4 b1_n;
    uint2 eh_n, el_n;
    f32x4 g_n, g2_n;
    auto fetch_rows = [&](const Idx& ix) {
        const int64_t e = ix.e, m = ix.m >= 0 ? ix.m : ix.e;
        b1_n = *reinterpret_cast<const u32x4*>(es + e * 8 + 4 * (g & 1));
        eh_n = *reinterpret_cast<const uint2*>(es + e * 8 + 2 * (g & 1));
        el_n = *reinterpret_cast<const uint2*>(es + e * 8 + 4 + 2 * (g & 1));
        if constexpr (S % 4 == 0) {
            const int qq = q0 < S ? q0 : 0;
            g_n = *reinterpret_cast<const f32x4*>(gout + e * S + qq);
            g2_n = *reinterpret_cast<const f32x4*>(gout + m * S + qq);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + r < S ? q0 + r : S - 1;    // clamped: always a readable element (masked in take)
                g_n[r] = gout[e * S + qq];
                g2_n[r] = gout[m * S + qq];
            }
        }
    };
    float gq[4], ye[4];
    bf16x8 B1, BE;
    auto take = [&](const Idx& ix) {
        asm volatile("" : "+v"(b1_n), "+v"(eh_n), "+v"(el_n), "+v"(g_n), "+v"(g2_n));
        B1 = __builtin_bit_cast(bf16x8, b1_n);
        BE = __builtin_bit_cast(bf16x8, u32x4{eh_n.x, eh_n.y, el_n.x, el_n.y});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gm = ix.m >= 0 ? g2_n[r] : 0.f;
            gq[r] = (ix.ok && q0 + r < S) ? g_n[r] + gm : 0.f;
        }
    };
    Idx ix_c = fetch_idx(t), ix_n = fetch_idx(t + stride);
    fetch_rows(ix_c);
    take(ix_c);
    for (; t < ntiles; t += stride) {
        GmlChainT T;
        fetch_rows(ix_n);                                      // next tile's rows are in flight while this one is computed
        const Idx ix_nn = fetch_idx(t + 2 * stride);
        __builtin_amdgcn_sched_barrier(0);
        gml_chain_forward<S, true>(W, T, B1, g);
        {
            const f32x4 ef = GML_MFMA(aE, BE, zero);          // rows 8..15 (lane groups 2, 3): e = hi + lo, exact
#pragma unroll
            for (int r = 0; r < 4; ++r) ye[r] = ef[r];
        }
        f32x4 go;
#pragma unroll
        for (int r = 0; r < 4; ++r) go[r] = (T.out[r] > 0.f) ? gq[r] : 0.f;
        const bf16x8 B3 = __builtin_bit_cast(bf16x8, gml_split_one(W.negI, go));      // [go hi | go lo]
        const f32x4 dh1 = GML_MFMA(a3[0], B3, zero);
        const f32x4 dh23 = GML_MFMA(a3[1], B3, zero);
        f32x4 gz1, gz2, gz3, y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gz1[r] = (T.z1[r] > 0.f) ? dh1[r] : 0.f;
            gz2[r] = dh23[r] * T.t3[r] * fmaf(-T.t2[r], T.t2[r], 1.f);
            gz3[r] = dh23[r] * T.t2[r] * fmaf(-T.t3[r], T.t3[r], 1.f);
            y[r] = (g < 2) ? go[r] : ye[r];
        }
        u32x4 g12h, g12l, g3yh, g3yl;
        gml_split_pair(W.negI, gz1, gz2, g12h, g12l);
        gml_split_pair(W.negI, gz3, y, g3yh, g3yl);
        bf16x8 XT[5], YTb;
        {
            const u32x4 im[6] = {T.hh, T.hl, g12h, g12l, g3yh, g3yl};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                *reinterpret_cast<uint2*>(trw + (4 * i + 0) * 512 + tr_wo) = uint2{im[2 * i].x, im[2 * i].y};
                *reinterpret_cast<uint2*>(trw + (4 * i + 1) * 512 + tr_wo) = uint2{im[2 * i + 1].x, im[2 * i + 1].y};
                *reinterpret_cast<uint2*>(trw + (4 * i + 2) * 512 + tr_wo) = uint2{im[2 * i].z, im[2 * i].w};
                *reinterpret_cast<uint2*>(trw + (4 * i + 3) * 512 + tr_wo) = uint2{im[2 * i + 1].z, im[2 * i + 1].w};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int b = 0; b < 5; ++b) XT[b] = gml_tr_frag(trw + (2 * b) * 512 + tr_ro, trw + (2 * b + 1) * 512 + tr_ro);
            YTb = gml_tr_frag(trw + 10 * 512 + tr_ro, trw + 11 * 512 + tr_ro);
            __builtin_amdgcn_wave_barrier();
        }
        const u32x4 YT = __builtin_bit_cast(u32x4, YTb);
        const bf16x8 Bhh = gml_op(YT.x, YT.y, YT.x, YT.y);
        const bf16x8 Bl0 = gml_op(YT.z, YT.w, 0u, 0u);
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            acc[b] = GML_MFMA(XT[b], Bl0, acc[b]);
            acc[b] = GML_MFMA(XT[b], Bhh, acc[b]);
        }
        __builtin_amdgcn_sched_barrier(0);
        take(ix_n);
        ix_n = ix_nn;
    }

    __syncthreads();
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][4 * b + r][lane] = acc[b][r];
    __syncthreads();
    float* P = partial + (int64_t)blockIdx.x * GML_CHAIN_NW(S);
    for (int it = threadIdx.x; it < 20 * 64; it += 256) {
        const int br = it >> 6, ln = it & 63;
        const float v = ((red[0][br][ln] + red[1][br][ln]) + red[2][br][ln]) + red[3][br][ln];
        const int b = br >> 2, row = 4 * (ln >> 4) + (br & 3), col = ln & 15;
        if (row >= H2) continue;
        if (b < 2) {
            if (col < S) P[6 * S * S + col * H4 + b * H2 + row] = v;
        } else {
            if (col >= 8 && col < 8 + S) P[(b - 2) * H2 * S + row * S + (col - 8)] = v;
        }
    }
}
