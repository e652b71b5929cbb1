// out[i][j] = sum_r A[r][i] . B[r][j] for a TALL pair of row-major matrices: A [N, a] (a up to thousands), B [N, b <= 128], N in the
// hundred thousands -- the weight gradient dW = Hcat^T g of the dense-block SpectConv (reference: /root/reference/libs/layers_tf.py:231-236,
// its autograd w.r.t. the weights; MNIST-75: a = S Fin = 768, b = Fout = 128, N = 307,200 at 4,096 graphs), which rounds 2-5 gave to a
// library GEMM in fp32 (435 us for the third layer: 139 TFLOP/s of the 157 the f32 matrix instruction has).
//
// bf16x3 split products on the bf16 matrix cores, fp32 accumulate -- the arithmetic of every other projection of this library.  The
// contraction runs over ROWS, so both operands must reach the MFMA with rows along K: a workgroup (8 waves) takes a 128-column tile of A
// and a range of 128-row groups; per group every thread loads its row's pieces (row = wave * 16 + lane & 15; 8 consecutive columns of
// each 32-column slab), splits them and writes row-major bf16 (hi, lo) images [slab][position][32 channels] into LDS (one ds_write_b128
// per image and slab, XOR-keyed rows: the scheme of the fused backward's dW phase, gml_spectconv_bwd3_impl.h); the waves read them
// TRANSPOSED (ds_read_b64_tr_b16) as A / B fragments; the tile's 8 x (b / 16) output blocks are dealt to the waves as 2 x 4 sub-grids
// (8 accumulator tiles and 6 fragment pairs per K step each).  One partial [a, b] per row range, folded in order (gml_fold_many).
// The column tiles of one row range run side by side (tile index fastest in the grid): B's rows are re-read out of the L2 / MALL.
#include "gml_common.h"

__host__ __device__ __forceinline__ int xw_tkey(int pos) { return ((pos >> 2) & 1) | ((((pos >> 1) ^ (pos >> 2) ^ (pos >> 3)) & 1) << 1); }

struct GmlXtyWideParams {
    const float* A; int64_t lda; const float* B; int64_t ldb;
    float* part; int64_t nrows; int32_t a, b, ntiles, nsplit; int64_t rows_per_split;
};

#define XW_ROWS 128
#define XW_IMG (XW_ROWS * 64)                                   /* bytes of one (hi or lo) image of one 32-column slab */

__global__ __launch_bounds__(512, 2) void gml_k_xty_wide(const GmlXtyWideParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* aimg = lds;                                   // [4 slabs][hi, lo][128 positions][64 bytes]
    unsigned char* bimg = lds + 8 * XW_IMG;                      // [4 slabs][hi, lo][128][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, kq = lane >> 4;
    const int ct = blockIdx.x % p.ntiles, sp = blockIdx.x / p.ntiles;
    const int64_t r_begin = (int64_t)sp * p.rows_per_split, r_end = min(r_begin + p.rows_per_split, p.nrows);
    const int c0 = 128 * ct;                                     // first A column of the tile
    const int nbs = (p.b + 31) / 32;                             // B slabs (<= 4)
    const int pos = wave * 16 + r16;
    const int woff = pos * 64 + (((kq ^ xw_tkey(pos)) & 3) << 4);
    const bool avec = (p.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
    const bool bvec = (p.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // transposing-read offsets of K step 0: lane (t = r16, kq): position 8 kq + 4 h + (t >> 2), 8-byte chunk 4 blk + (t & 3)
    const int tj = r16 >> 2, tc = r16 & 3;
    auto roffs = [&](int h, int blk) {
        const int ps = 8 * kq + 4 * h + tj, cidx = 4 * blk + tc;
        return ps * 64 + ((((cidx >> 1) ^ xw_tkey(ps)) & 3) << 4) + ((cidx & 1) << 3);
    };
    // the 8 x (b / 16) grid of 16 x 16 output blocks of the tile is dealt to the 8 waves as sub-grids (fewer fragment reads than one A
    // block against every B block: the kernel is bound by its LDS reads): wide B (b > 64): A slab `wave & 3` (its 2 blocks) x B blocks
    // 4 (wave >> 2) .. + 3;  b <= 64: A block (slab wave & 3, half wave >> 2) x all (<= 4) B blocks
    const bool wide = nbs > 2;
    const int sa = wave & 3, wh = wave >> 2;
    const int off16[2][2] = {{roffs(0, 0), roffs(1, 0)}, {roffs(0, 1), roffs(1, 1)}};   // [16-column half of a slab][h]

    auto load8 = [&](const float* base, int64_t ld, int64_t row, int col, int ncols, bool vec, float (&v)[8]) {
        const float* q = base + row * ld + col;
        const bool rv = row < r_end;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (rv && vec && col + 4 * h + 4 <= ncols) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(q + 4 * h);
                v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[4 * h + u] = (rv && col + 4 * h + u < ncols) ? q[4 * h + u] : 0.f;
            }
        }
    };

    // B slabs beyond b are never written by the group loop: zero them once (their blocks are computed and dropped, not stored)
    for (int i = tid; i < (4 - nbs) * 2 * XW_IMG / 16; i += 512)
        *reinterpret_cast<f32x4*>(bimg + 2 * nbs * XW_IMG + 16 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
    // the next group's row pieces travel in registers while this group's contraction runs (one workgroup per CU: nothing else hides
    // the loads' latency)
    float va[4][8], vb[4][8];
    auto fetch = [&](int64_t g0) {
        const int64_t row = g0 + pos;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            load8(p.A, p.lda, row, c0 + 32 * s + 8 * kq, p.a, avec, va[s]);
            if (s < nbs) load8(p.B, p.ldb, row, 32 * s + 8 * kq, p.b, bvec, vb[s]);
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (int64_t g0 = r_begin; g0 < r_end; g0 += XW_ROWS) {
        // ---- the thread's row pieces -> bf16 (hi, lo) -> images
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 h, l;
            gml_split8(va[s], h, l);
            *reinterpret_cast<bf16x8*>(aimg + (2 * s) * XW_IMG + woff) = h;
            *reinterpret_cast<bf16x8*>(aimg + (2 * s + 1) * XW_IMG + woff) = l;
            if (s < nbs) {
                gml_split8(vb[s], h, l);
                *reinterpret_cast<bf16x8*>(bimg + (2 * s) * XW_IMG + woff) = h;
                *reinterpret_cast<bf16x8*>(bimg + (2 * s + 1) * XW_IMG + woff) = l;
            }
        }
        __syncthreads();
        if (g0 + XW_ROWS < r_end) fetch(g0 + XW_ROWS);
        // ---- contraction over the group's 128 rows: 4 K steps of 32 positions
#pragma unroll
        for (int st = 0; st < XW_ROWS / 32; ++st) {
            const unsigned char* xa = aimg + (2 * sa) * XW_IMG + st * 2048;
            if (wide) {
                bf16x8 fah[2], fal[2], fbh[4], fbl[4];
#pragma unroll
                for (int fb = 0; fb < 2; ++fb) {
                    fah[fb] = gml_tr_frag(xa + off16[fb][0], xa + off16[fb][1]);
                    fal[fb] = gml_tr_frag(xa + XW_IMG + off16[fb][0], xa + XW_IMG + off16[fb][1]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {                    // B block 4 wh + i: slab 2 wh + (i >> 1), half i & 1
                    const unsigned char* xb = bimg + (2 * (2 * wh + (i >> 1))) * XW_IMG + st * 2048;
                    fbh[i] = gml_tr_frag(xb + off16[i & 1][0], xb + off16[i & 1][1]);
                    fbl[i] = gml_tr_frag(xb + XW_IMG + off16[i & 1][0], xb + XW_IMG + off16[i & 1][1]);
                }
#pragma unroll
                for (int fb = 0; fb < 2; ++fb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[4 * fb + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[fb], fbh[i], acc[4 * fb + i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[4 * fb + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[fb], fbl[i], acc[4 * fb + i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[4 * fb + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[fb], fbh[i], acc[4 * fb + i], 0, 0, 0);
                }
            } else {
                const bf16x8 fah = gml_tr_frag(xa + off16[wh][0], xa + off16[wh][1]);
                const bf16x8 fal = gml_tr_frag(xa + XW_IMG + off16[wh][0], xa + XW_IMG + off16[wh][1]);
                bf16x8 fbh[4], fbl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned char* xb = bimg + (2 * (i >> 1)) * XW_IMG + st * 2048;
                    fbh[i] = gml_tr_frag(xb + off16[i & 1][0], xb + off16[i & 1][1]);
                    fbl[i] = gml_tr_frag(xb + XW_IMG + off16[i & 1][0], xb + XW_IMG + off16[i & 1][1]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal, fbh[i], acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah, fbl[i], acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah, fbh[i], acc[i], 0, 0, 0);
            }
        }
        __syncthreads();                                         // images free for the next group
    }
    // ---- partial of this row range: D[i = A column 4 kq + reg of the block][j = B column r16 of the block]
    float* out = p.part + (int64_t)sp * p.a * p.b;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (!wide && q >= 4) break;
        const int fb = wide ? (q >> 2) : wh, jb = wide ? 4 * wh + (q & 3) : q;
        const int j = 16 * jb + r16;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = c0 + 32 * sa + 16 * fb + 4 * kq + reg;
            if (i < p.a && j < p.b) out[(int64_t)i * p.b + j] = acc[q][reg];
        }
    }
}

static int xw_splits(int64_t n, int ntiles) {
    int64_t s = (2 * GML_NUM_CU) / ntiles;                       // two workgroups per CU (128 KB... one resident; the second queues behind it)
    if (s < 1) s = 1;
    const int64_t groups = gml_cdiv(n, XW_ROWS);
    if (s > groups) s = groups;
    return (int)(s < 1 ? 1 : s);
}

// 1 when gml_xty_wide takes the shape (b <= 128; a <= 4096)
extern "C" int gml_xty_wide_supported(int64_t n, int32_t a, int32_t b) { return (n >= 0 && a >= 1 && a <= 4096 && b >= 1 && b <= 128) ? 1 : 0; }

extern "C" size_t gml_xty_wide_workspace_bytes(int64_t n, int32_t a, int32_t b) {
    if (!gml_xty_wide_supported(n, a, b) || n == 0) return 0;
    return (size_t)xw_splits(n, (a + 127) / 128) * (size_t)a * b * sizeof(float);
}

extern "C" int gml_xty_wide(const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t n, int32_t a, int32_t b,
                            void* ws, size_t ws_bytes, gml_stream_t stream) {
    if (!gml_xty_wide_supported(n, a, b)) return GML_E_UNSUPPORTED;
    if (!out || lda < a || ldb < b) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { gml_zero_async(out, sizeof(float) * (size_t)a * b, st); return gml_launch_status(); }
    if (!A || !B || !ws || ws_bytes < gml_xty_wide_workspace_bytes(n, a, b)) return GML_E_WORKSPACE;
    GmlXtyWideParams p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.part = (float*)ws; p.nrows = n; p.a = a; p.b = b;
    p.ntiles = (a + 127) / 128;
    p.nsplit = xw_splits(n, p.ntiles);
    p.rows_per_split = (gml_cdiv(gml_cdiv(n, XW_ROWS), p.nsplit)) * XW_ROWS;
    const size_t lds = 16 * XW_IMG;
    GML_ALLOW_BIG_LDS(rc, (&gml_k_xty_wide), 160 * 1024)
    if (rc != hipSuccess) return (int)rc;
    hipLaunchKernelGGL(gml_k_xty_wide, dim3((unsigned)(p.ntiles * p.nsplit)), dim3(512), lds, st, p);
    const int lrc = gml_launch_status();
    if (lrc != GML_OK) return lrc;
    gml_fold_job job = {};
    job.partial = (const float*)ws; job.nparts = p.nsplit; job.n = (int64_t)a * b; job.dst[0] = out; job.ndst[0] = (int64_t)a * b;
    return gml_fold_many(&job, 1, stream);
}
