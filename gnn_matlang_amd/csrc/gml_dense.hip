// Dense-block SpectConv for batches of equal-size graphs with near-dense masks (MNIST-75: the TF reference formulates the
// layer exactly this way, /root/reference/libs/layers_tf.py:231-236:  s0 = matmul(support[:, i], x);  out += s0 . W_i).
//
// One kernel, both directions -- the batched support product
//
//      out[b n + r][s so + f]  (=, or summed over s)   sum_k  D[b][s][r][k] . act[b n + k][s sa + f]
//
//   forward :  D = support blocks as stored (row = target node), act = X (sa = 0: the same tile for every support),
//              out = Hcat [B n, S Fin] (so = Fin), whose product with the stacked weights is one tall GEMM;
//   backward:  D = the transposed blocks, act = d Hcat (sa = Fin), the S products summed in registers -> d X.
//
// Machine mapping: one workgroup per graph, one wave per 16 support rows (n = 75 -> 5 waves).  The supports are per
// data set constants, kept in HBM as bf16 (hi, lo) images [B][S][2][n][KP] (gml_dense_pack: x = hi + lo to 2^-17, rows
// padded to KP = 32 ceil(n / 32) with zeros): a lane's MFMA operand (its row, 8 consecutive k) is ONE 16-byte load straight
// from HBM, no LDS, nothing shared between waves, every support byte read once.  The activation tile goes through LDS
// once per graph (forward) or once per support (backward): fp32 -> (hi, lo) row-major bf16 images [k][f], read back
// transposed by ds_read_b64_tr_b16 as the other operand.  The product is computed transposed (D[i = f][j = row]): a lane
// ends up with 4 consecutive features of its own row = one 16-byte store.  bf16x3: hi.hi + lo.hi + hi.lo in fp32
// accumulators (relative error ~2^-16 per product; the tolerance of the parity tests is 1e-4).
#include "gml_common.h"

typedef short dn_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) dn_s16x4 dn_lds_s16x4;

struct GmlDenseParams {
    const uint16_t* dimg;
    const float* act;
    float* out;
    int64_t lda, ldo;
    int32_t sa, so, B, S, n, KP, F, vec_in, vec_out;
};

__device__ __forceinline__ uint32_t dn_pack2(float a, float b) {             // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// one MFMA operand (8 k-slots) from two transposing reads (lane semantics probed by tools/probes/probe_tr.hip)
__device__ __forceinline__ bf16x8 dn_tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const dn_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dn_lds_s16x4*)(p0));
    const dn_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dn_lds_s16x4*)(p1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
#define DN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// image row pitch in bytes: 32 bytes per 16-feature tile + 16 (tools/lds_sim.py: the transposing reads of 8 rows x 32 bytes
// then fall on distinct banks; a single tile needs no pad)
__host__ __device__ constexpr int dn_pitch(int nft) { return 32 * nft + (nft == 1 ? 0 : 16); }

template <int NFT, bool ACC>
__global__ __launch_bounds__(384) void gml_k_dense_support_mm(GmlDenseParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dn_lds[];
    constexpr int PA = dn_pitch(NFT);
    constexpr int NCH = 4 * NFT;                             // 8-byte chunks (4 features) per image row
    constexpr int KSMAX = 3;                                 // KP <= 96
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int t16 = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x, n = p.n, KP = p.KP, KS = KP >> 5, F = p.F;
    unsigned char* img_h = dn_lds;
    unsigned char* img_l = dn_lds + KP * PA;
    const int row = wave * 16 + t16;                         // this lane's support row (the column of the transposed product)
    const int rowc = row < n ? row : n - 1;
    const float* actb = p.act + (int64_t)b * n * p.lda;
    float* outr = p.out + ((int64_t)b * n + rowc) * p.ldo;

    auto stage = [&](int s) {                                // act[:, s sa : s sa + F] of this graph -> (hi, lo) images [k][f]
        const float* a = actb + s * p.sa;
        for (int idx = tid; idx < KP * NCH; idx += nthr) {
            const int k = idx / NCH, ch = idx % NCH;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < n && 4 * ch < F) {
                const float* q = a + (int64_t)k * p.lda + 4 * ch;
                if (p.vec_in) v = *reinterpret_cast<const f32x4*>(q);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (4 * ch + j < F) v[j] = q[j];
                }
            }
            const uint32_t h0 = dn_pack2(v[0], v[1]), h1 = dn_pack2(v[2], v[3]);
            const uint32_t l0 = dn_pack2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
            const uint32_t l1 = dn_pack2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
            *reinterpret_cast<uint2*>(img_h + k * PA + 8 * ch) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(img_l + k * PA + 8 * ch) = uint2{l0, l1};
        }
    };
    // support operand of this lane for support s: row rowc, k = 32 ks + 8 kq .. + 7 of the hi and the lo image
    u32x4 bh[KSMAX], bl[KSMAX];
    auto load_rows = [&](int s) {
        const uint16_t* base = p.dimg + ((int64_t)(b * p.S + s) * 2 * n + rowc) * KP + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int kc = ks < KS ? ks : KS - 1;            // (clamped: the loads stay unconditional, the product skips ks >= KS)
            bh[ks] = *reinterpret_cast<const u32x4*>(base + 32 * kc);
            bl[ks] = *reinterpret_cast<const u32x4*>(base + (int64_t)n * KP + 32 * kc);
        }
    };
    // transposing-read addresses: lane (t, kq) passes row 8 kq + (t >> 2) (+ 4 for the second read), 8-byte chunk (t & 3)
    const int aoff = (8 * kq + (t16 >> 2)) * PA + 8 * (t16 & 3);

    f32x4 acc[NFT];
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto store = [&](int s) {
        if (row >= n) return;
        float* o = outr + s * p.so + 4 * kq;
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            const int f0 = 16 * ft + 4 * kq;
            if (f0 >= F) continue;
            if (p.vec_out) *reinterpret_cast<f32x4*>(o + 16 * ft) = acc[ft];
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (f0 + j < F) o[16 * ft + j] = acc[ft][j];
            }
        }
    };

    load_rows(0);
    for (int s = 0; s < p.S; ++s) {
        if (s == 0 || p.sa != 0) {
            if (s > 0) __syncthreads();                      // every wave is done with the previous tile
            stage(s);
            __syncthreads();
        }
        u32x4 ch[KSMAX], cl[KSMAX];
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) { ch[ks] = bh[ks]; cl[ks] = bl[ks]; }
        if (s + 1 < p.S) load_rows(s + 1);                   // next support's rows in flight during this product
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            if (ks < KS) {
                const bf16x8 Bh = __builtin_bit_cast(bf16x8, ch[ks]), Bl = __builtin_bit_cast(bf16x8, cl[ks]);
                const unsigned char* ah = img_h + 32 * ks * PA + aoff;
                const unsigned char* al = img_l + 32 * ks * PA + aoff;
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) {
                    const bf16x8 Ah = dn_tr_frag(ah + 32 * ft, ah + 32 * ft + 4 * PA);
                    const bf16x8 Al = dn_tr_frag(al + 32 * ft, al + 32 * ft + 4 * PA);
                    acc[ft] = DN_MFMA(Al, Bh, acc[ft]);
                    acc[ft] = DN_MFMA(Ah, Bl, acc[ft]);
                    acc[ft] = DN_MFMA(Ah, Bh, acc[ft]);
                }
            }
        }
        if constexpr (!ACC) {
            store(s);
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if constexpr (ACC) store(0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Chained forward (libs/layers_tf.py:231-236 does `matmul(support, x)` THEN `tensordot(., W_i)` in one
// op sequence): out = sum_s (D_s X) W_s + bias with the support product's accumulators as the projection's operand -- Hcat
// [B n, S Fin] (230 KB per graph at MNIST's third layer) is neither written nor re-read by a library GEMM on this path
// (WRITE_H: it is still written when the caller wants it for the weight gradient H^T g).
//
// The support product leaves (D_s X)^T in MFMA D layout: lane (row j, kq) holds features 16 ft + 4 kq + reg.  The k-slot
// order of an MFMA is free as long as both operands use the same one, so the projection's K = 32 step t takes slot (kq, i)
// = feature 32 t + 4 kq + i (i < 4) | 32 t + 16 + 4 kq + i - 4 (i >= 4): exactly what the lane already holds of tiles 2 t and
// 2 t + 1 -- split to bf16 (hi, lo), no data movement.  The weights come pre-arranged in that order by gml_dense_pack_w:
// wimg[s][t][ot][hi | lo][lane][8] (lane (o = 16 ot + (lane & 15), kq)), one 16-byte load per lane and fragment from L2
// (Wcat 768 x 128 as bf16 hi / lo = 393 KB).  out^T = W^T H^T again lands transposed: a lane stores 4 consecutive outputs of
// its own row.  bf16x3 throughout.
// Work items of the projection: (support s, group q of DN_ITEM_T K = 32 steps); a ring of three LDS buffers receives the items'
// weight fragments by LDS-DMA from DN_NLD loader waves, two items ahead of their use.
template <int NFT>
struct DnChain {
    static constexpr int KT = (NFT + 1) / 2;                 // K = 32 steps of the projection
    static constexpr int ITEM_T = KT >= 4 ? 2 : KT;          // steps per work item
    static constexpr int NQ = KT / ITEM_T;                   // items per support
};
#define DN_NLD 3

__device__ __forceinline__ void dn_dma16(u32x4 rs, uint32_t lds_addr, int voff) {      // (see gml_dma16, gml_spectconv_fwd3_impl.h)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}

template <int NFT, int NOT, bool WRITE_H>
__global__ __launch_bounds__(576) void gml_k_dense_conv_fwd(GmlDenseParams p, const uint16_t* __restrict__ wimg,
                                                            const float* __restrict__ bias, float* __restrict__ out2,
                                                            int64_t ldo2, int Fout, int relu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dn_lds[];
    using CH = DnChain<NFT>;
    constexpr int PA = dn_pitch(NFT);
    constexpr int NCH = 4 * NFT;
    constexpr int KSMAX = 3;
    constexpr int KT = CH::KT, ITEM_T = CH::ITEM_T, NQ = CH::NQ;
    constexpr int ITEM_BYTES = ITEM_T * NOT * 2 * 64 * 16;   // fragments of one work item: [t][ot][hi | lo][lane][16 bytes]
    constexpr int NINST = ITEM_BYTES / 1024;                 // DMA instructions per item
    static_assert(NINST % DN_NLD == 0 || NINST < DN_NLD || true, "");
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncw = (int)(blockDim.x >> 6) - DN_NLD;         // compute waves (one per 16 support rows)
    const int t16 = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x, n = p.n, KP = p.KP, KS = KP >> 5, F = p.F;
    unsigned char* img_h = dn_lds;
    unsigned char* img_l = dn_lds + KP * PA;
    unsigned char* ring = dn_lds + 2 * KP * PA;              // 3 x ITEM_BYTES
    const int nitems = p.S * NQ;
    const float* actb = p.act + (int64_t)b * n * p.lda;

    for (int idx = tid; idx < KP * NCH; idx += nthr) {       // X of this graph -> (hi, lo) images [k][f], once (every wave helps)
        const int k = idx / NCH, ch = idx % NCH;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k < n && 4 * ch < F) {
            const float* q = actb + (int64_t)k * p.lda + 4 * ch;
            if (p.vec_in) v = *reinterpret_cast<const f32x4*>(q);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (4 * ch + j < F) v[j] = q[j];
            }
        }
        const uint32_t h0 = dn_pack2(v[0], v[1]), h1 = dn_pack2(v[2], v[3]);
        const uint32_t l0 = dn_pack2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
        const uint32_t l1 = dn_pack2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
        *reinterpret_cast<uint2*>(img_h + k * PA + 8 * ch) = uint2{h0, h1};
        *reinterpret_cast<uint2*>(img_l + k * PA + 8 * ch) = uint2{l0, l1};
    }

    if (wave >= ncw) {
        // ---- loader waves: item i -> ring[i % 3], issued two items ahead; (barrier i) = item i landed and everyone left item i - 1
        const int li = wave - ncw;
        const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)dn_lds) + 2 * KP * PA;
        const uint64_t wa = reinterpret_cast<uint64_t>(wimg);
        const u32x4 rs = u32x4{(uint32_t)wa, (uint32_t)(wa >> 32) & 0xffffu, (uint32_t)(p.S * KT * NOT * 2 * 64 * 16), 0x00020000u};
        constexpr int CW = (NINST + DN_NLD - 1) / DN_NLD;    // instructions per loader and item (a static count: the waits rely on it)
        auto issue = [&](int i) {
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const int j = min(li + DN_NLD * c, NINST - 1);   // (the last ones repeat a piece: the count stays static)
                dn_dma16(rs, lds0 + (i % 3) * ITEM_BYTES + j * 1024, i * ITEM_BYTES + j * 1024 + lane * 16);
            }
        };
        issue(0);
        if (nitems > 1) issue(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // (A) X images complete
        for (int i = 0; i < nitems; ++i) {
            if (i + 1 < nitems) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CW) : "memory");   // item i landed (item i + 1 may be in flight)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // (B i)
            asm volatile("" ::: "memory");
            if (i + 2 < nitems) issue(i + 2);
        }
        return;
    }

    // ---- compute waves
    const int row = wave * 16 + t16;
    const int rowc = row < n ? row : n - 1;
    float* outr = p.out + ((int64_t)b * n + rowc) * p.ldo;
    u32x4 bh[KSMAX], bl[KSMAX];
    auto load_rows = [&](int s) {
        const uint16_t* base = p.dimg + ((int64_t)(b * p.S + s) * 2 * n + rowc) * KP + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int kc = ks < KS ? ks : KS - 1;
            bh[ks] = *reinterpret_cast<const u32x4*>(base + 32 * kc);
            bl[ks] = *reinterpret_cast<const u32x4*>(base + (int64_t)n * KP + 32 * kc);
        }
    };
    const int aoff = (8 * kq + (t16 >> 2)) * PA + 8 * (t16 & 3);
    f32x4 oacc[NOT];
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) oacc[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_rows(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // (A)
    asm volatile("" ::: "memory");
    int item = 0;
    for (int s = 0; s < p.S; ++s) {
        f32x4 acc[NFT];
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 ch[KSMAX], cl[KSMAX];
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) { ch[ks] = bh[ks]; cl[ks] = bl[ks]; }
        if (s + 1 < p.S) load_rows(s + 1);
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            if (ks < KS) {
                const bf16x8 Bh = __builtin_bit_cast(bf16x8, ch[ks]), Bl = __builtin_bit_cast(bf16x8, cl[ks]);
                const unsigned char* ah = img_h + 32 * ks * PA + aoff;
                const unsigned char* al = img_l + 32 * ks * PA + aoff;
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) {
                    const bf16x8 Ah = dn_tr_frag(ah + 32 * ft, ah + 32 * ft + 4 * PA);
                    const bf16x8 Al = dn_tr_frag(al + 32 * ft, al + 32 * ft + 4 * PA);
                    acc[ft] = DN_MFMA(Al, Bh, acc[ft]);
                    acc[ft] = DN_MFMA(Ah, Bl, acc[ft]);
                    acc[ft] = DN_MFMA(Ah, Bh, acc[ft]);
                }
            }
        }
        if constexpr (WRITE_H) {                              // Hcat for the weight gradient (see the header)
            if (row < n) {
                float* o = outr + s * p.so + 4 * kq;
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) {
                    const int f0 = 16 * ft + 4 * kq;
                    if (f0 >= F) continue;
                    if (p.vec_out) *reinterpret_cast<f32x4*>(o + 16 * ft) = acc[ft];
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (f0 + j < F) o[16 * ft + j] = acc[ft][j];
                    }
                }
            }
        }
        // ---- projection of this support: out^T += W_s^T (D_s X)^T, one ring buffer per group of ITEM_T K steps
#pragma unroll
        for (int q = 0; q < NQ; ++q, ++item) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // (B item): fragments landed; every wave has left the previous item
            asm volatile("" ::: "memory");
            const unsigned char* wb = ring + (item % 3) * ITEM_BYTES + lane * 16;
#pragma unroll
            for (int tt = 0; tt < ITEM_T; ++tt) {
                const int t = q * ITEM_T + tt;
                const float hv[8] = {acc[2 * t][0], acc[2 * t][1], acc[2 * t][2], acc[2 * t][3],
                                     (2 * t + 1 < NFT) ? acc[(2 * t + 1 < NFT) ? 2 * t + 1 : 0][0] : 0.f,
                                     (2 * t + 1 < NFT) ? acc[(2 * t + 1 < NFT) ? 2 * t + 1 : 0][1] : 0.f,
                                     (2 * t + 1 < NFT) ? acc[(2 * t + 1 < NFT) ? 2 * t + 1 : 0][2] : 0.f,
                                     (2 * t + 1 < NFT) ? acc[(2 * t + 1 < NFT) ? 2 * t + 1 : 0][3] : 0.f};
                bf16x8 hh, hl;
                gml_split8(hv, hh, hl);
#pragma unroll
                for (int ot = 0; ot < NOT; ++ot) {
                    const unsigned char* wq = wb + (tt * NOT + ot) * 2 * 64 * 16;
                    const bf16x8 Wh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wq));
                    const bf16x8 Wl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wq + 64 * 16));
                    oacc[ot] = DN_MFMA(Wl, hh, oacc[ot]);
                    oacc[ot] = DN_MFMA(Wh, hl, oacc[ot]);
                    oacc[ot] = DN_MFMA(Wh, hh, oacc[ot]);
                }
            }
        }
    }
    if (row < n) {
        float* o = out2 + ((int64_t)b * n + row) * ldo2 + 4 * kq;
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot) {
            const int o0 = 16 * ot + 4 * kq;
            if (o0 >= Fout) continue;
            f32x4 v = oacc[ot];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (bias != nullptr && o0 + j < Fout) v[j] += bias[o0 + j];
                if (relu) v[j] = fmaxf(v[j], 0.f);
            }
            if (Fout % 4 == 0 && ldo2 % 4 == 0) *reinterpret_cast<f32x4*>(o + 16 * ot) = v;
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (o0 + j < Fout) o[16 * ot + j] = v[j];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Chained backward for dX = sum_s D_s^T (G W_s^T)   (autograd of the layer above w.r.t. x): the projection G W_s^T runs per
// wave on its own 16 rows (B operand = the lane's own G row, 8 consecutive o per K step straight from HBM, split to bf16 once
// per graph; A = W_s fragments [f rows][o slots] from the loader waves' ring), its result -- (G W_s^T)^T in D layout, what the
// forward's support product produced -- goes into the LDS images, and the support product with the TRANSPOSED blocks
// contracts over the graph's rows.  d Hcat [B n, S Fin] is neither written nor read.
template <int NFT, int NOT>
__global__ __launch_bounds__(576) void gml_k_dense_conv_bwdx(GmlDenseParams p, const uint16_t* __restrict__ wimgT,
                                                             int64_t ldg, int Fout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dn_lds[];
    constexpr int PA = dn_pitch(NFT);
    constexpr int KSMAX = 3;
    constexpr int KT2 = NOT / 2;                             // K = 32 steps over o
    constexpr int FQ = NFT >= 8 ? 2 : 1;                     // work items per support (groups of NFT / FQ feature tiles)
    constexpr int FT_I = NFT / FQ;
    constexpr int ITEM_BYTES = FT_I * KT2 * 2 * 64 * 16;     // [ft][t][hi | lo][lane][16 bytes]
    constexpr int NINST = ITEM_BYTES / 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncw = (int)(blockDim.x >> 6) - DN_NLD;
    const int t16 = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x, n = p.n, KP = p.KP, KS = KP >> 5, F = p.F;
    unsigned char* img_h = dn_lds;
    unsigned char* img_l = dn_lds + KP * PA;
    unsigned char* ring = dn_lds + 2 * KP * PA;
    const int nitems = p.S * FQ;
    // image rows n .. KP - 1 (the K padding of the product) are zero and stay so: only rows < n are ever written
    for (int idx = tid; idx < 2 * KP * PA / 16; idx += blockDim.x) reinterpret_cast<u32x4*>(dn_lds)[idx] = u32x4{0u, 0u, 0u, 0u};

    if (wave >= ncw) {
        const int li = wave - ncw;
        const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)dn_lds) + 2 * KP * PA;
        const uint64_t wa = reinterpret_cast<uint64_t>(wimgT);
        const u32x4 rs = u32x4{(uint32_t)wa, (uint32_t)(wa >> 32) & 0xffffu, (uint32_t)(p.S * NFT * KT2 * 2 * 64 * 16), 0x00020000u};
        constexpr int CW = (NINST + DN_NLD - 1) / DN_NLD;
        auto issue = [&](int i) {
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const int j = min(li + DN_NLD * c, NINST - 1);
                dn_dma16(rs, lds0 + (i % 3) * ITEM_BYTES + j * 1024, i * ITEM_BYTES + j * 1024 + lane * 16);
            }
        };
        issue(0);
        if (nitems > 1) issue(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // (A) images zeroed
        int i = 0;
        for (int s = 0; s < p.S; ++s) {
            for (int q = 0; q < FQ; ++q, ++i) {
                if (i + 1 < nitems) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                // (B i)
                asm volatile("" ::: "memory");
                if (i + 2 < nitems) issue(i + 2);
            }
            __builtin_amdgcn_s_barrier();                    // (C s) images of support s complete
            __builtin_amdgcn_s_barrier();                    // (D s) product of support s done in every wave
        }
        return;
    }

    const int row = wave * 16 + t16;
    const int rowc = row < n ? row : n - 1;
    // the lane's own G row, split once: K step t covers o = 32 t + 8 kq .. + 7
    bf16x8 gh[KT2], gl[KT2];
    {
        const float* gr = p.act + ((int64_t)b * n + rowc) * ldg;
#pragma unroll
        for (int t = 0; t < KT2; ++t) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int o = 32 * t + 8 * kq + j;
                v[j] = (row < n && o < Fout) ? gr[o] : 0.f;
            }
            gml_split8(v, gh[t], gl[t]);
        }
    }
    u32x4 bh[KSMAX], bl[KSMAX];
    auto load_rows = [&](int s) {
        const uint16_t* base = p.dimg + ((int64_t)(b * p.S + s) * 2 * n + rowc) * KP + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int kc = ks < KS ? ks : KS - 1;
            bh[ks] = *reinterpret_cast<const u32x4*>(base + 32 * kc);
            bl[ks] = *reinterpret_cast<const u32x4*>(base + (int64_t)n * KP + 32 * kc);
        }
    };
    const int aoff = (8 * kq + (t16 >> 2)) * PA + 8 * (t16 & 3);
    f32x4 dxacc[NFT];
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) dxacc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_rows(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // (A)
    asm volatile("" ::: "memory");
    int item = 0;
    for (int s = 0; s < p.S; ++s) {
        // ---- T_s^T = W_s G^T for the wave's rows: D[i = f][j = row]
        f32x4 tacc[NFT];
#pragma unroll
        for (int q = 0; q < FQ; ++q, ++item) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // (B item)
            asm volatile("" ::: "memory");
            const unsigned char* wb = ring + (item % 3) * ITEM_BYTES + lane * 16;
#pragma unroll
            for (int fi = 0; fi < FT_I; ++fi) {
                f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < KT2; ++t) {
                    const unsigned char* wq = wb + (fi * KT2 + t) * 2 * 64 * 16;
                    const bf16x8 Wh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wq));
                    const bf16x8 Wl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wq + 64 * 16));
                    d = DN_MFMA(Wl, gh[t], d);
                    d = DN_MFMA(Wh, gl[t], d);
                    d = DN_MFMA(Wh, gh[t], d);
                }
                tacc[q * FT_I + fi] = d;
            }
        }
        // ---- the wave's rows of T_s -> (hi, lo) images [k = row][f]
        if (row < n) {
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) {
                const f32x4 v = tacc[ft];
                const uint32_t h0 = dn_pack2(v[0], v[1]), h1 = dn_pack2(v[2], v[3]);
                const uint32_t l0 = dn_pack2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
                const uint32_t l1 = dn_pack2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
                *reinterpret_cast<uint2*>(img_h + row * PA + 32 * ft + 8 * kq) = uint2{h0, h1};
                *reinterpret_cast<uint2*>(img_l + row * PA + 32 * ft + 8 * kq) = uint2{l0, l1};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // (C s)
        asm volatile("" ::: "memory");
        // ---- dX^T += T_s^T D_s: the support product on the transposed blocks
        u32x4 ch[KSMAX], cl[KSMAX];
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) { ch[ks] = bh[ks]; cl[ks] = bl[ks]; }
        if (s + 1 < p.S) load_rows(s + 1);
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            if (ks < KS) {
                const bf16x8 Bh = __builtin_bit_cast(bf16x8, ch[ks]), Bl = __builtin_bit_cast(bf16x8, cl[ks]);
                const unsigned char* ah = img_h + 32 * ks * PA + aoff;
                const unsigned char* al = img_l + 32 * ks * PA + aoff;
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) {
                    const bf16x8 Ah = dn_tr_frag(ah + 32 * ft, ah + 32 * ft + 4 * PA);
                    const bf16x8 Al = dn_tr_frag(al + 32 * ft, al + 32 * ft + 4 * PA);
                    dxacc[ft] = DN_MFMA(Al, Bh, dxacc[ft]);
                    dxacc[ft] = DN_MFMA(Ah, Bl, dxacc[ft]);
                    dxacc[ft] = DN_MFMA(Ah, Bh, dxacc[ft]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // (D s)
        asm volatile("" ::: "memory");
    }
    if (row < n) {
        float* o = p.out + ((int64_t)b * n + row) * p.ldo + 4 * kq;
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            const int f0 = 16 * ft + 4 * kq;
            if (f0 >= F) continue;
            if (p.vec_out) *reinterpret_cast<f32x4*>(o + 16 * ft) = dxacc[ft];
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (f0 + j < F) o[16 * ft + j] = dxacc[ft][j];
            }
        }
    }
}

// weight [S][Fin][Fout] -> fragments of A = W_s [f rows][o slots] for the chained backward: wimgT[s][ft][t][hi | lo][lane][8],
// lane (f = 16 ft + (lane & 15), kq), slot j: o = 32 t + 8 kq + j
__global__ __launch_bounds__(256) void gml_k_dense_pack_wT(const float* __restrict__ w, uint16_t* __restrict__ wimg, int S, int Fin,
                                                           int Fout, int NFT, int KT2) {
    const int64_t total = (int64_t)S * NFT * KT2 * 64 * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int slot = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const int64_t blk = i >> 9;                          // (s * NFT + ft) * KT2 + t
        const int t = (int)(blk % KT2), ft = (int)((blk / KT2) % NFT), s = (int)(blk / ((int64_t)KT2 * NFT));
        const int f = 16 * ft + (lane & 15), o = 32 * t + 8 * (lane >> 4) + slot;
        const float v = (f < Fin && o < Fout) ? w[((int64_t)s * Fin + f) * Fout + o] : 0.f;
        const uint32_t h = dn_pack2(v, 0.f) & 0xffffu;
        const uint32_t l = dn_pack2(v - __uint_as_float(h << 16), 0.f) & 0xffffu;
        wimg[((blk * 2) * 64 + lane) * 8 + slot] = (uint16_t)h;
        wimg[((blk * 2 + 1) * 64 + lane) * 8 + slot] = (uint16_t)l;
    }
}

static void dn_bwd_shape(int Fin, int Fout, int& NFT, int& NOT) {
    const int nft = (Fin + 15) / 16;
    NFT = nft <= 1 ? 1 : (nft <= 2 ? 2 : (nft <= 4 ? 4 : 8));
    NOT = Fout <= 64 ? 4 : 8;
}

extern "C" size_t gml_dense_wimgt_elems(int32_t S, int32_t Fin, int32_t Fout) {
    int NFT, NOT;
    dn_bwd_shape(Fin, Fout, NFT, NOT);
    return (size_t)S * NFT * (NOT / 2) * 2 * 64 * 8;
}

extern "C" int gml_dense_pack_wt(const float* w, uint16_t* wimg, int32_t S, int32_t Fin, int32_t Fout, void* stream) {
    if (w == nullptr || wimg == nullptr) return GML_E_BADARG;
    if (S < 1 || Fin < 1 || Fin > 128 || Fout < 1 || Fout > 128) return GML_E_UNSUPPORTED;
    int NFT, NOT;
    dn_bwd_shape(Fin, Fout, NFT, NOT);
    const int64_t total = (int64_t)S * NFT * (NOT / 2) * 64 * 8;
    hipLaunchKernelGGL(gml_k_dense_pack_wT, dim3((unsigned)gml_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wimg, S, Fin,
                       Fout, NFT, NOT / 2);
    return gml_launch_status();
}

template <int NFT, int NOT>
static int dn_launch_bwdx(const GmlDenseParams& p, const uint16_t* wimgT, int64_t ldg, int Fout, hipStream_t st) {
    constexpr int FQ = NFT >= 8 ? 2 : 1;
    const size_t lds = (size_t)2 * p.KP * dn_pitch(NFT) + (size_t)3 * (NFT / FQ) * (NOT / 2) * 2 * 64 * 16;
    GML_ALLOW_BIG_LDS(rc, (gml_k_dense_conv_bwdx<NFT, NOT>), 160 * 1024);
    if (rc != hipSuccess) return (int)rc;
    const int nwaves = (p.n + 15) / 16;
    hipLaunchKernelGGL((gml_k_dense_conv_bwdx<NFT, NOT>), dim3((unsigned)p.B), dim3(64 * (nwaves + DN_NLD)), lds, st, p, wimgT, ldg, Fout);
    return gml_launch_status();
}

// dx[(b n + i) lddx + f] = sum_s sum_j D[b][s][j][i] (sum_o g[(b n + j) ldg + o] W[s][f][o]);  dimgT: the packed TRANSPOSED blocks
// (gml_dense_pack(transpose = 1)), wimgT from gml_dense_pack_wt.  Fin, Fout <= 128.
extern "C" int gml_dense_conv_bwd_x(const uint16_t* dimgT, const float* g, int64_t ldg, const uint16_t* wimgT, float* dx, int64_t lddx,
                                    int32_t B, int32_t S, int32_t n, int32_t KP, int32_t Fin, int32_t Fout, void* stream) {
    if (dimgT == nullptr || g == nullptr || wimgT == nullptr || dx == nullptr || ldg < Fout || lddx < Fin) return GML_E_BADARG;
    if (n < 1 || n > 96 || KP % 32 != 0 || KP < n || KP > 96 || Fin < 1 || Fin > 128 || Fout < 1 || Fout > 128 || S < 1 || B < 0)
        return GML_E_UNSUPPORTED;
    if (B == 0) return GML_OK;
    GmlDenseParams p;
    p.dimg = dimgT; p.act = g; p.out = dx; p.lda = ldg; p.ldo = lddx; p.sa = 0; p.so = 0;
    p.B = B; p.S = S; p.n = n; p.KP = KP; p.F = Fin;
    p.vec_in = 0;
    p.vec_out = (Fin % 4 == 0 && lddx % 4 == 0 && ((uintptr_t)dx & 15) == 0) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    int NFT, NOT;
    dn_bwd_shape(Fin, Fout, NFT, NOT);
#define DN_BWDX(A) return NOT == 4 ? dn_launch_bwdx<A, 4>(p, wimgT, ldg, Fout, st) : dn_launch_bwdx<A, 8>(p, wimgT, ldg, Fout, st)
    if (NFT == 1) { DN_BWDX(1); }
    if (NFT == 2) { DN_BWDX(2); }
    if (NFT == 4) { DN_BWDX(4); }
    DN_BWDX(8);
#undef DN_BWDX
}

// weight [S][Fin][Fout] fp32 -> projection fragments wimg[s][t][ot][hi | lo][lane][8] in the k-slot order of the chained kernel
__global__ __launch_bounds__(256) void gml_k_dense_pack_w(const float* __restrict__ w, uint16_t* __restrict__ wimg, int S, int Fin,
                                                          int Fout, int KT, int NOT) {
    const int64_t total = (int64_t)S * KT * NOT * 64 * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int slot = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const int64_t blk = i >> 9;                          // (s * KT + t) * NOT + ot
        const int ot = (int)(blk % NOT), t = (int)((blk / NOT) % KT), s = (int)(blk / ((int64_t)NOT * KT));
        const int o = 16 * ot + (lane & 15), kq = lane >> 4;
        const int f = 32 * t + (slot < 4 ? 4 * kq + slot : 16 + 4 * kq + slot - 4);
        const float v = (f < Fin && o < Fout) ? w[((int64_t)s * Fin + f) * Fout + o] : 0.f;
        const uint32_t h = dn_pack2(v, 0.f) & 0xffffu;
        const uint32_t l = dn_pack2(v - __uint_as_float(h << 16), 0.f) & 0xffffu;
        wimg[((blk * 2) * 64 + lane) * 8 + slot] = (uint16_t)h;
        wimg[((blk * 2 + 1) * 64 + lane) * 8 + slot] = (uint16_t)l;
    }
}

extern "C" size_t gml_dense_wimg_elems(int32_t S, int32_t Fin, int32_t Fout) {
    const int KT = ((Fin + 15) / 16 + 1) / 2, NOT = Fout <= 64 ? 4 : 8;
    return (size_t)S * KT * NOT * 2 * 64 * 8;
}

extern "C" int gml_dense_pack_w(const float* w, uint16_t* wimg, int32_t S, int32_t Fin, int32_t Fout, void* stream) {
    if (w == nullptr || wimg == nullptr) return GML_E_BADARG;
    if (S < 1 || Fin < 1 || Fin > 128 || Fout < 1 || Fout > 128) return GML_E_UNSUPPORTED;
    const int nft = (Fin + 15) / 16, NFT = nft <= 1 ? 1 : (nft <= 2 ? 2 : (nft <= 4 ? 4 : 8));
    const int KT = (NFT + 1) / 2, NOT = Fout <= 64 ? 4 : 8;
    const int64_t total = (int64_t)S * KT * NOT * 64 * 8;
    hipLaunchKernelGGL(gml_k_dense_pack_w, dim3((unsigned)gml_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wimg, S, Fin,
                       Fout, KT, NOT);
    return gml_launch_status();
}

template <int NFT, int NOT>
static int dn_launch_conv(const GmlDenseParams& p, const uint16_t* wimg, const float* bias, float* out2, int64_t ldo2, int Fout,
                          int relu, bool write_h, hipStream_t st) {
    const size_t lds = (size_t)2 * p.KP * dn_pitch(NFT) + (size_t)3 * DnChain<NFT>::ITEM_T * NOT * 2 * 64 * 16;
    {
        GML_ALLOW_BIG_LDS(rc1, (gml_k_dense_conv_fwd<NFT, NOT, true>), 160 * 1024);
        GML_ALLOW_BIG_LDS(rc0, (gml_k_dense_conv_fwd<NFT, NOT, false>), 160 * 1024);
        if (rc1 != hipSuccess) return (int)rc1;
        if (rc0 != hipSuccess) return (int)rc0;
    }
    const int nwaves = (p.n + 15) / 16;
    if (write_h) hipLaunchKernelGGL((gml_k_dense_conv_fwd<NFT, NOT, true>), dim3((unsigned)p.B), dim3(64 * (nwaves + DN_NLD)), lds, st, p, wimg, bias, out2, ldo2, Fout, relu);
    else hipLaunchKernelGGL((gml_k_dense_conv_fwd<NFT, NOT, false>), dim3((unsigned)p.B), dim3(64 * (nwaves + DN_NLD)), lds, st, p, wimg, bias, out2, ldo2, Fout, relu);
    return gml_launch_status();
}

// out2[(b n + r) ldo2 + o] = act?( sum_s sum_f (sum_k D[b][s][r][k] x[(b n + k) ldx + f]) W[s][f][o] + bias[o] );  hcat (optional):
// [B n, S Fin] receives D_s X as gml_dense_support_mm(sum_s = 0) would write it.  wimg from gml_dense_pack_w.  Fin, Fout <= 128.
extern "C" int gml_dense_conv_fwd(const uint16_t* dimg, const float* x, int64_t ldx, const uint16_t* wimg, const float* bias,
                                  float* out2, int64_t ldo2, float* hcat, int32_t B, int32_t S, int32_t n, int32_t KP, int32_t Fin,
                                  int32_t Fout, int32_t relu, void* stream) {
    if (dimg == nullptr || x == nullptr || wimg == nullptr || out2 == nullptr || ldx < Fin || ldo2 < Fout) return GML_E_BADARG;
    if (n < 1 || n > 96 || KP % 32 != 0 || KP < n || KP > 96 || Fin < 1 || Fin > 128 || Fout < 1 || Fout > 128 || S < 1 || B < 0)
        return GML_E_UNSUPPORTED;
    if (B == 0) return GML_OK;
    GmlDenseParams p;
    p.dimg = dimg; p.act = x; p.out = hcat; p.lda = ldx; p.ldo = (int64_t)S * Fin; p.sa = 0; p.so = Fin;
    p.B = B; p.S = S; p.n = n; p.KP = KP; p.F = Fin;
    p.vec_in = (Fin % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0) ? 1 : 0;
    p.vec_out = (hcat != nullptr && Fin % 4 == 0 && ((uintptr_t)hcat & 15) == 0) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const int nft = (Fin + 15) / 16;
    const bool wh = hcat != nullptr;
#define DN_CONV(NFT)                                                                                             \
    return Fout <= 64 ? dn_launch_conv<NFT, 4>(p, wimg, bias, out2, ldo2, Fout, relu, wh, st)                     \
                      : dn_launch_conv<NFT, 8>(p, wimg, bias, out2, ldo2, Fout, relu, wh, st)
    if (nft <= 1) { DN_CONV(1); }
    if (nft <= 2) { DN_CONV(2); }
    if (nft <= 4) { DN_CONV(4); }
    DN_CONV(8);
#undef DN_CONV
}

// fp32 blocks [B][S][n][n] (row-major; transpose = 1 takes block^T) -> bf16 (hi, lo) images [B][S][2][n][KP]
__global__ __launch_bounds__(256) void gml_k_dense_pack(const float* __restrict__ blocks, uint16_t* __restrict__ img,
                                                        int64_t nblocks, int n, int KP, int transpose) {
    const int64_t total = nblocks * n * KP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % KP);
        const int r = (int)((i / KP) % n);
        const int64_t blk = i / ((int64_t)KP * n);
        float v = 0.f;
        if (k < n) v = blocks[blk * n * n + (transpose ? (int64_t)k * n + r : (int64_t)r * n + k)];
        const uint32_t h = dn_pack2(v, 0.f) & 0xffffu;
        const float res = v - __uint_as_float(h << 16);
        const uint32_t l = dn_pack2(res, 0.f) & 0xffffu;
        img[(blk * 2) * n * KP + (int64_t)r * KP + k] = (uint16_t)h;
        img[(blk * 2 + 1) * n * KP + (int64_t)r * KP + k] = (uint16_t)l;
    }
}

extern "C" int gml_dense_pack(const float* blocks, uint16_t* img, int64_t nblocks, int32_t n, int32_t KP, int32_t transpose,
                              void* stream) {
    if (blocks == nullptr || img == nullptr) return GML_E_BADARG;
    if (n < 1 || n > 96 || KP % 32 != 0 || KP < n || KP > 96 || nblocks < 0) return GML_E_UNSUPPORTED;
    if (nblocks == 0) return GML_OK;
    const int64_t total = nblocks * n * KP;
    int64_t grid = gml_cdiv(total, 256);
    if (grid > 64 * GML_NUM_CU) grid = 64 * GML_NUM_CU;
    hipLaunchKernelGGL(gml_k_dense_pack, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, blocks, img, nblocks, n, KP,
                       transpose);
    return gml_launch_status();
}

template <int NFT, bool ACC>
static int dn_launch(const GmlDenseParams& p, hipStream_t st) {
    const size_t lds = (size_t)2 * p.KP * dn_pitch(NFT);
    GML_ALLOW_BIG_LDS(rc, (gml_k_dense_support_mm<NFT, ACC>), 2 * 96 * dn_pitch(NFT));   // (the attribute latches per device: the largest n)
    if (rc != hipSuccess) return (int)rc;
    const int nwaves = (p.n + 15) / 16;
    hipLaunchKernelGGL((gml_k_dense_support_mm<NFT, ACC>), dim3((unsigned)p.B), dim3(64 * nwaves), lds, st, p);
    return gml_launch_status();
}

// out[(b n + r) ldo + s so + f] (=, or += over s when sum_s) sum_k D[b][s][r][k] act[(b n + k) lda + s sa + f],  f < F <= 128
extern "C" int gml_dense_support_mm(const uint16_t* dimg, const float* act, int64_t lda, int32_t sa, float* out, int64_t ldo,
                                    int32_t so, int32_t sum_s, int32_t B, int32_t S, int32_t n, int32_t KP, int32_t F,
                                    void* stream) {
    if (dimg == nullptr || act == nullptr || out == nullptr || lda < F || ldo < F) return GML_E_BADARG;
    if (n < 1 || n > 96 || KP % 32 != 0 || KP < n || KP > 96 || F < 1 || F > 128 || S < 1 || B < 0) return GML_E_UNSUPPORTED;
    if (B == 0) return GML_OK;
    GmlDenseParams p;
    p.dimg = dimg; p.act = act; p.out = out; p.lda = lda; p.ldo = ldo; p.sa = sa; p.so = so;
    p.B = B; p.S = S; p.n = n; p.KP = KP; p.F = F;
    p.vec_in = (F % 4 == 0 && lda % 4 == 0 && sa % 4 == 0 && ((uintptr_t)act & 15) == 0) ? 1 : 0;
    p.vec_out = (F % 4 == 0 && ldo % 4 == 0 && so % 4 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const int nft = (F + 15) / 16;
#define DN_CASE(NFT)                                                              \
    return sum_s ? dn_launch<NFT, true>(p, st) : dn_launch<NFT, false>(p, st)
    if (nft <= 1) { DN_CASE(1); }
    if (nft <= 2) { DN_CASE(2); }
    if (nft <= 4) { DN_CASE(4); }
    DN_CASE(8);
#undef DN_CASE
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the dense-block layer WITHOUT Hcat and without a library GEMM (the TF reference's
// autograd of libs/layers_tf.py:231-236):
//
//      dW[s][f][o] = sum_b sum_j H_s[b][j][f] g[b n + j][o],      H_s[b] = D[b][s] X[b]     (recomputed per graph, never stored)
//
// Rounds 2-4 wrote Hcat [B n, S Fin] in the forward (0.94 GB per step at MNIST's third layer and 4,096 graphs), read it back as a
// row-slab batched library GEMM and summed the slabs.  Here a workgroup owns (support s, a block of 64 output columns, a slice of
// the graphs): per graph the X tile goes through LDS as in the support product; a wave forms H for its 16 rows UNtransposed --
// D[i = row j][n = feature]: lane (f = lane & 15, kq) holds rows j = 4 kq + reg -- which is, after the bf16 (hi, lo) split, exactly
// the A operand (i = f, k = j = 4 kq + reg) of a K = 16 MFMA; the B operand g[j][o] of the same rows is 4 strided dwords per lane
// and output tile.  bf16x3 both times.  The (feature tile, output tile) accumulators stay in registers across the slice's graphs;
// the waves' row blocks are summed through LDS in fixed order at the end; one partial per slice -> gml_fold_many.
// The support product is recomputed once per output-column block (2 blocks at Fout = 128): 2 x 360 of the kernel's ~1,700 MFMA
// equivalents per graph and support -- the price of keeping 6 x 128 x 128 accumulators out of one workgroup's registers.
typedef short dn_s16x4v __attribute__((ext_vector_type(4)));

template <int NFT, int NOTB>
__global__ __launch_bounds__(384) void gml_k_dense_dw(const uint16_t* __restrict__ dimg, const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ g, int64_t ldg, float* __restrict__ partial,
                                                     int B, int S, int n, int KP, int Fin, int Fout, int per_slice, int vec_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dn_lds[];
    constexpr int PA = dn_pitch(NFT), NCH = 4 * NFT, KSMAX = 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int t16 = lane & 15, kq = lane >> 4;
    const int KS = KP >> 5;
    const int slice = blockIdx.x, s = blockIdx.y, ob0 = blockIdx.z * NOTB;   // first 16-column output tile of this workgroup
    unsigned char* img_h = dn_lds;
    unsigned char* img_l = dn_lds + KP * PA;
    const int row = wave * 16 + t16, rowc = row < n ? row : n - 1;
    const int aoff = (8 * kq + (t16 >> 2)) * PA + 8 * (t16 & 3);
    f32x4 acc[NFT][NOTB];
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int ot = 0; ot < NOTB; ++ot) acc[ft][ot] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int b0 = slice * per_slice, b1 = min(b0 + per_slice, B);
    for (int b = b0; b < b1; ++b) {
        if (b > b0) __syncthreads();                           // every wave is done with the previous graph's X images
        const float* xb = x + (int64_t)b * n * ldx;
        for (int idx = tid; idx < KP * NCH; idx += nthr) {     // X of this graph -> (hi, lo) images [k][f]
            const int k = idx / NCH, ch = idx % NCH;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < n && 4 * ch < Fin) {
                const float* q = xb + (int64_t)k * ldx + 4 * ch;
                if (vec_in) v = *reinterpret_cast<const f32x4*>(q);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (4 * ch + j < Fin) v[j] = q[j];
                }
            }
            const uint32_t h0 = dn_pack2(v[0], v[1]), h1 = dn_pack2(v[2], v[3]);
            const uint32_t l0 = dn_pack2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
            const uint32_t l1 = dn_pack2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
            *reinterpret_cast<uint2*>(img_h + k * PA + 8 * ch) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(img_l + k * PA + 8 * ch) = uint2{l0, l1};
        }
        // this lane's support row (row rowc, k = 32 ks + 8 kq .. + 7, hi and lo image) and its g values: rows 16 wave + 4 kq + r, column o
        u32x4 bh[KSMAX], bl[KSMAX];
        const uint16_t* base = dimg + ((int64_t)(b * S + s) * 2 * n + rowc) * KP + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int kc = ks < KS ? ks : KS - 1;
            bh[ks] = *reinterpret_cast<const u32x4*>(base + 32 * kc);
            bl[ks] = *reinterpret_cast<const u32x4*>(base + (int64_t)n * KP + 32 * kc);
        }
        dn_s16x4v gh[NOTB], gl[NOTB];
#pragma unroll
        for (int ot = 0; ot < NOTB; ++ot) {
            const int o = 16 * (ob0 + ot) + t16;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * wave + 4 * kq + r;
                v[r] = (j < n && o < Fout) ? g[((int64_t)b * n + j) * ldg + o] : 0.f;     // rows past n contribute nothing
            }
            const uint32_t h0 = dn_pack2(v[0], v[1]), h1 = dn_pack2(v[2], v[3]);
            const uint32_t l0 = dn_pack2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
            const uint32_t l1 = dn_pack2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
            gh[ot] = __builtin_bit_cast(dn_s16x4v, uint2{h0, h1});
            gl[ot] = __builtin_bit_cast(dn_s16x4v, uint2{l0, l1});
        }
        __syncthreads();
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            // H tile (rows of this wave x features 16 ft .. + 15), untransposed: A = the support rows, B = X^T fragments
            f32x4 h = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) {
                if (ks < KS) {
                    const bf16x8 Dh = __builtin_bit_cast(bf16x8, bh[ks]), Dl = __builtin_bit_cast(bf16x8, bl[ks]);
                    const unsigned char* ah = img_h + 32 * ks * PA + aoff + 32 * ft;
                    const unsigned char* al = img_l + 32 * ks * PA + aoff + 32 * ft;
                    const bf16x8 Xh = dn_tr_frag(ah, ah + 4 * PA);
                    const bf16x8 Xl = dn_tr_frag(al, al + 4 * PA);
                    h = DN_MFMA(Dl, Xh, h);
                    h = DN_MFMA(Dh, Xl, h);
                    h = DN_MFMA(Dh, Xh, h);
                }
            }
            // lane (f = 16 ft + t16, kq) holds H[j = 16 wave + 4 kq + reg][f]: the A operand (i = f, k = j) of the K = 16 contraction
            const uint32_t h0 = dn_pack2(h[0], h[1]), h1 = dn_pack2(h[2], h[3]);
            const uint32_t l0 = dn_pack2(h[0] - __uint_as_float(h0 << 16), h[1] - __uint_as_float(h0 & 0xffff0000u));
            const uint32_t l1 = dn_pack2(h[2] - __uint_as_float(h1 << 16), h[3] - __uint_as_float(h1 & 0xffff0000u));
            const dn_s16x4v Hh = __builtin_bit_cast(dn_s16x4v, uint2{h0, h1}), Hl = __builtin_bit_cast(dn_s16x4v, uint2{l0, l1});
#pragma unroll
            for (int ot = 0; ot < NOTB; ++ot) {
                acc[ft][ot] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Hl, gh[ot], acc[ft][ot], 0, 0, 0);
                acc[ft][ot] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Hh, gl[ot], acc[ft][ot], 0, 0, 0);
                acc[ft][ot] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Hh, gh[ot], acc[ft][ot], 0, 0, 0);
            }
        }
    }
    // ---- the waves' row blocks summed in fixed order, one feature tile at a time through red[wave][ot][lane] (f32x4; NOTB KB per wave:
    //      all tiles at once would be 160 KB at Fin = 128 -- one workgroup per CU), then one partial per slice
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(dn_lds);
    const int nw = nthr >> 6;
    float* P = partial + ((int64_t)slice * S + s) * Fin * Fout;
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) {
        if (ft > 0) __syncthreads();
#pragma unroll
        for (int ot = 0; ot < NOTB; ++ot) red[(wave * NOTB + ot) * 64 + lane] = acc[ft][ot];
        __syncthreads();
        for (int idx = tid; idx < NOTB * 64; idx += nthr) {
            const int ot = idx >> 6, ln = idx & 63;
            f32x4 t = red[ot * 64 + ln];
            for (int w = 1; w < nw; ++w) t += red[(w * NOTB + ot) * 64 + ln];
            // D layout of the contraction: lane (o = ln & 15, kq = ln >> 4) holds features 16 ft + 4 kq + reg
            const int o = 16 * (ob0 + ot) + (ln & 15);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int f = 16 * ft + 4 * (ln >> 4) + reg;
                if (f < Fin && o < Fout) P[(int64_t)f * Fout + o] = t[reg];
            }
        }
    }
}

extern "C" int32_t gml_dense_dw_slices(int32_t B) {
    int s = B < 64 ? B : 64;
    return s < 1 ? 1 : s;
}
extern "C" size_t gml_dense_dw_workspace_bytes(int32_t B, int32_t S, int32_t Fin, int32_t Fout) {
    if (B <= 0 || S <= 0 || Fin <= 0 || Fout <= 0) return 0;
    return (size_t)gml_dense_dw_slices(B) * S * Fin * Fout * sizeof(float);
}

template <int NFT>
static int dn_launch_dw(const uint16_t* dimg, const float* x, int64_t ldx, const float* g, int64_t ldg, float* partial, int B, int S, int n,
                        int KP, int Fin, int Fout, hipStream_t st) {
    constexpr int NOTB = 4;                                    // (2 at Fin = 128 -- a second workgroup per CU -- measured slower: twice the recomputed support products)
    const int slices = gml_dense_dw_slices(B), per = (B + slices - 1) / slices;
    const int nwaves = (n + 15) / 16;
    size_t lds = (size_t)2 * KP * dn_pitch(NFT);
    const size_t red = (size_t)nwaves * NOTB * 64 * 16;
    if (red > lds) lds = red;
    if (lds > 160 * 1024) return GML_E_UNSUPPORTED;
    GML_ALLOW_BIG_LDS(rc, (gml_k_dense_dw<NFT, NOTB>), 160 * 1024);
    if (rc != hipSuccess) return (int)rc;
    const int vec_in = (Fin % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL((gml_k_dense_dw<NFT, NOTB>), dim3((unsigned)slices, (unsigned)S, (unsigned)((Fout + 16 * NOTB - 1) / (16 * NOTB))),
                       dim3(64 * nwaves), lds, st, dimg, x, ldx, g, ldg, partial, B, S, n, KP, Fin, Fout, per, vec_in);
    return gml_launch_status();
}

// dW[s][f][o] = sum over the batch of (D[b][s] X[b])^T g[b]: the partial sums of gml_dense_dw_slices(B) graph slices into ws
// ([slices][S][Fin][Fout]) and their fold in slice order into dw (when dw != NULL; NULL: the partials stay for gml_fold_many).
extern "C" int gml_dense_conv_bwd_w(const uint16_t* dimg, const float* x, int64_t ldx, const float* g, int64_t ldg, float* dw,
                                    int32_t B, int32_t S, int32_t n, int32_t KP, int32_t Fin, int32_t Fout, void* ws, size_t ws_bytes,
                                    void* stream) {
    if (!dimg || !x || !g || ldx < Fin || ldg < Fout) return GML_E_BADARG;
    if (n < 1 || n > 96 || KP % 32 != 0 || KP < n || KP > 96 || Fin < 1 || Fin > 128 || Fout < 1 || S < 1 || B < 0) return GML_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nw = (int64_t)S * Fin * Fout;
    if (B == 0) { if (dw) gml_zero_async(dw, sizeof(float) * nw, st); return gml_launch_status(); }
    if (!ws || ws_bytes < gml_dense_dw_workspace_bytes(B, S, Fin, Fout)) return GML_E_WORKSPACE;
    const int nft = (Fin + 15) / 16;
    int rc;
    if (nft <= 1) rc = dn_launch_dw<1>(dimg, x, ldx, g, ldg, (float*)ws, B, S, n, KP, Fin, Fout, st);
    else if (nft <= 2) rc = dn_launch_dw<2>(dimg, x, ldx, g, ldg, (float*)ws, B, S, n, KP, Fin, Fout, st);
    else if (nft <= 4) rc = dn_launch_dw<4>(dimg, x, ldx, g, ldg, (float*)ws, B, S, n, KP, Fin, Fout, st);
    else rc = dn_launch_dw<8>(dimg, x, ldx, g, ldg, (float*)ws, B, S, n, KP, Fin, Fout, st);
    if (rc != GML_OK || !dw) return rc;
    gml_fold_job job = {};
    job.partial = (const float*)ws; job.nparts = gml_dense_dw_slices(B); job.n = nw; job.dst[0] = dw; job.ndst[0] = nw;
    return gml_fold_many(&job, 1, stream);
}
