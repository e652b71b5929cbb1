// Readout head + L1-sum loss of the ZINC GNNML3 for LARGE batches (reference: /root/reference/Zinc12k.py:343-345 head, :365 loss):
//
//   h = relu(W1 p + b1),   pre = w2 . h + b2,   loss = sum_r valid[r] |pre_r - y_r|          p = pooled features [R, 32], 32 hidden units
//
// gml_head.hip does this in one workgroup for the reference's batch of 64.  At the bench's 131,072 graphs per step the general road
// was ~16 launches (two library GEMMs + relu + sub / abs / sum forward; sign, three GEMMs, two tall X^T Y products, two column sums,
// a relu mask backward) for 16 MB of pooled rows: 0.2 ms of a 9 ms step.  Here: ONE pass forward (a lane owns a row: 32 x 32 FMAs
// against W1 broadcast from LDS; per-workgroup partial of the loss) and ONE pass backward (the row's forward recomputed, d loss / d p
// written, the weight gradients  dW1 = dh^T P, db1 = dh^T 1, dw2 = (s h)^T 1, db2 = s^T 1  contracted over the rows on the matrix
// cores -- v_mfma_f32_16x16x4_f32, exact fp32 products, each wave its own 64 rows, as in gml_k_ml3_split_bwd -- one partial per
// workgroup), each followed by the fixed-order fold of the partials.  Every sum has a fixed order: bitwise repeatable.
#include "gml_common.h"

#define HB_ROWS 256                 /* rows per tile = threads per workgroup */
#define HB_W 32                     /* nin = nh = 32 (GNNML3's 30 + 2 features, fc1: 32 -> 32) */
#define HB_LDP (HB_W + 1)           /* B tile rows [p | .]: odd stride, row-per-lane accesses conflict free */
#define HB_LDA (2 * HB_W + 1 + 2)   /* A tile rows [dh (32) | s h (32) | s | pad]: 67 floats */
#define HB_NPART (HB_W * HB_W + HB_W + HB_W + 1)   /* dw1 | db1 | dw2 | db2 */

struct GmlHeadBigParams {
    const float* p; int64_t ldp;
    const float* y; const float* valid;                      // [Rl]; valid may be NULL (all ones)
    const float* w1; const float* b1; const float* w2; const float* b2;
    int64_t R, Rl;                                           // pooled rows, rows that enter the loss (<= R: the others get a zero gradient)
    int32_t ntiles;
    float* part;                                             // forward: [grid] loss partials; backward: [grid][HB_NPART]
    const float* gscale;                                     // backward: upstream gradient of the loss (device scalar; NULL = 1)
    float* gp; int64_t ldgp;
};

// the row's hidden activations and logit: W1 rows broadcast from LDS
__device__ __forceinline__ float hb_forward(const float (&pr)[HB_W], const float* w1s, const float* b1s, const float* w2s, float b2,
                                            float (&h)[HB_W]) {
    float pre = b2;
#pragma unroll
    for (int o = 0; o < HB_W; ++o) {
        float a = b1s[o];
#pragma unroll
        for (int k = 0; k < HB_W / 4; ++k) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(w1s + o * HB_W + 4 * k);
            a = fmaf(pr[4 * k], w.x, a); a = fmaf(pr[4 * k + 1], w.y, a); a = fmaf(pr[4 * k + 2], w.z, a); a = fmaf(pr[4 * k + 3], w.w, a);
        }
        // (left alone, the LDS reads of all 32 independent dot products are scheduled in front of the arithmetic: ~620 spilled
        //  registers.  The fence is tied to the unit's sum: its FMAs stay above it, the next unit's reads below)
        asm volatile("" : "+v"(a) :: "memory");
        h[o] = fmaxf(a, 0.f);
        pre = fmaf(w2s[o], h[o], pre);
    }
    return pre;
}

__device__ __forceinline__ void hb_load_weights(const GmlHeadBigParams& q, float* w1s, float* b1s, float* w2s) {
    for (int i = threadIdx.x; i < HB_W * HB_W / 4; i += HB_ROWS) *reinterpret_cast<f32x4*>(w1s + 4 * i) = *reinterpret_cast<const f32x4*>(q.w1 + 4 * i);
    if (threadIdx.x < HB_W) {
        b1s[threadIdx.x] = q.b1 ? q.b1[threadIdx.x] : 0.f;
        w2s[threadIdx.x] = q.w2[threadIdx.x];
    }
}

__device__ __forceinline__ void hb_load_row(const GmlHeadBigParams& q, int64_t r, float (&pr)[HB_W]) {
    const float* src = q.p + min(r, q.R - 1) * q.ldp;
#pragma unroll
    for (int k = 0; k < HB_W / 4; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * k);
        pr[4 * k] = t.x; pr[4 * k + 1] = t.y; pr[4 * k + 2] = t.z; pr[4 * k + 3] = t.w;
    }
}

__global__ __launch_bounds__(HB_ROWS) void gml_k_headbig_fwd(const GmlHeadBigParams q) {
    __shared__ __attribute__((aligned(16))) float w1s[HB_W * HB_W];
    __shared__ float b1s[HB_W], w2s[HB_W], red[HB_ROWS];
    hb_load_weights(q, w1s, b1s, w2s);
    __syncthreads();
    const float b2 = q.b2 ? q.b2[0] : 0.f;
    float acc = 0.f;
    for (int t = blockIdx.x; t < q.ntiles; t += gridDim.x) {
        const int64_t r = (int64_t)t * HB_ROWS + threadIdx.x;
        float pr[HB_W], h[HB_W];
        hb_load_row(q, r, pr);
        const float pre = hb_forward(pr, w1s, b1s, w2s, b2, h);
        if (r < q.Rl) acc += fabsf(pre - q.y[r]) * (q.valid ? q.valid[r] : 1.f);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = HB_ROWS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) q.part[blockIdx.x] = red[0];
}

// loss[0] = sum of the partials in ascending order (one workgroup: a fixed tree); loss_sum[0] += loss (optional)
__global__ __launch_bounds__(256) void gml_k_headbig_loss_fold(const float* __restrict__ part, int n, float* __restrict__ loss,
                                                               float* __restrict__ loss_sum) {
    __shared__ float red[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss[0] = red[0];
        if (loss_sum) loss_sum[0] += red[0];
    }
}

__global__ __launch_bounds__(HB_ROWS) void gml_k_headbig_bwd(const GmlHeadBigParams q) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* w1s = sm;                                         // [32][32]
    float* b1s = w1s + HB_W * HB_W;                          // [32]
    float* w2s = b1s + HB_W;                                 // [32]
    float* At = w2s + HB_W;                                  // [HB_ROWS][HB_LDA]   dh | s h | s
    float* Bt = At + HB_ROWS * HB_LDA;                       // [HB_ROWS][HB_LDP]   p
    hb_load_weights(q, w1s, b1s, w2s);
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const float b2 = q.b2 ? q.b2[0] : 0.f;
    const float gs = q.gscale ? q.gscale[0] : 1.f;
    // accumulators of this wave's rows: A column blocks cb 0, 1 (dh), 2, 3 (s h), 4 (s) x B blocks fb 0, 1 (p), 2 (ones column)
    f32x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = blockIdx.x; t < q.ntiles; t += gridDim.x) {
        const int64_t r = (int64_t)t * HB_ROWS + tid;
        float pr[HB_W], h[HB_W];
        hb_load_row(q, r, pr);
        const float* w1p = w1s;
        const float pre = hb_forward(pr, w1p, b1s, w2s, b2, h);
        float s = 0.f;
        if (r < q.Rl) {
            const float d = pre - q.y[r];
            s = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (q.valid ? q.valid[r] : 1.f) * gs;     // torch.sign: 0 at 0
        }
        // (the wave reads only its own rows of the tiles below, written by its own lanes: no workgroup barrier; the previous
        //  tile's fragment reads of this wave are behind it in program order)
        float* ar = At + tid * HB_LDA;
        float* br = Bt + tid * HB_LDP;
#pragma unroll
        for (int f = 0; f < HB_W; ++f) br[f] = (r < q.R) ? pr[f] : 0.f;
        float dh[HB_W];
#pragma unroll
        for (int o = 0; o < HB_W; ++o) {
            dh[o] = h[o] > 0.f ? s * w2s[o] : 0.f;
            ar[o] = dh[o];
            ar[HB_W + o] = s * h[o];
        }
        ar[2 * HB_W] = s;
        // d loss / d p of the row, four inputs at a time: gp[4 k ..] = sum_o dh[o] W1[o][4 k ..] (the fence is tied to the sums: the
        // next chunk's 32 LDS reads stay below it -- left free, all 256 are scheduled first and spill)
        float* dst = q.gp + min(r, q.R - 1) * q.ldgp;
#pragma unroll
        for (int k = 0; k < HB_W / 4; ++k) {
            f32x4 g4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < HB_W; ++o) g4 += dh[o] * *reinterpret_cast<const f32x4*>(w1p + o * HB_W + 4 * k);
            asm volatile("" : "+v"(g4) :: "memory");
            if (r < q.R) *reinterpret_cast<f32x4*>(dst + 4 * k) = g4;
        }
        // D[c][f] += sum_rows A[row][c] B[row][f] over this wave's 64 rows
        const int rb = wave * 64;
#pragma unroll 4
        for (int t16 = 0; t16 < 16; ++t16) {
            const int rr = rb + 4 * t16 + kq;
            const float* a_ = At + rr * HB_LDA;
            const float* b_ = Bt + rr * HB_LDP;
            const float a0 = a_[r16], a1 = a_[16 + r16], a2 = a_[32 + r16], a3 = a_[48 + r16], a4 = (r16 == 0) ? a_[64] : 0.f;
            const float b0 = b_[r16], b1 = b_[16 + r16], one = (r16 == 0) ? 1.f : 0.f;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, one, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, one, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, one, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, one, acc[7], 0, 0, 0);
            acc[8] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4, one, acc[8], 0, 0, 0);
        }
    }
    // ---- one partial per workgroup: the four waves folded in fixed order through LDS.  D layout: lane (j = r16, rows i = 4 kq + reg)
    __syncthreads();
    float* wred = At;                                        // [4 waves][9][4][64]
#pragma unroll
    for (int b = 0; b < 9; ++b)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) wred[((wave * 9 + b) * 4 + reg) * 64 + lane] = acc[b][reg];
    __syncthreads();
    float* P = q.part + (int64_t)blockIdx.x * HB_NPART;
    for (int it = tid; it < 9 * 4 * 64; it += HB_ROWS) {
        const int b = it / 256, reg = (it >> 6) & 3, ln = it & 63;
        float v = wred[((0 * 9 + b) * 4 + reg) * 64 + ln];
#pragma unroll
        for (int w = 1; w < HB_ROWS / 64; ++w) v += wred[((w * 9 + b) * 4 + reg) * 64 + ln];
        const int i = 4 * (ln >> 4) + reg, j = ln & 15;      // row i (A column inside the block), column j (B column inside the block)
        if (b < 4) {                                          // dw1[c][f]: c = 16 (b >> 1) + i, f = 16 (b & 1) + j
            P[(16 * (b >> 1) + i) * HB_W + 16 * (b & 1) + j] = v;
        } else if (j == 0) {
            if (b < 6) P[HB_W * HB_W + 16 * (b - 4) + i] = v;                        // db1
            else if (b < 8) P[HB_W * HB_W + HB_W + 16 * (b - 6) + i] = v;            // dw2
            else if (i == 0) P[HB_W * HB_W + 2 * HB_W] = v;                          // db2
        }
    }
}

// fold [nparts][HB_NPART] partials in fixed order into dw1 | db1 | dw2 | db2 (db1, db2 may be NULL)
__global__ __launch_bounds__(256) void gml_k_headbig_fold(const float* __restrict__ part, int64_t nparts, float* __restrict__ dw1,
                                                          float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    __shared__ float red[16][17];
    const int jl = threadIdx.x & 15, wl = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + jl;
    float a = 0.f;
    if (j < HB_NPART) a = gml_fold_column(part, nparts, HB_NPART, j, wl);
    red[wl][jl] = a;
    __syncthreads();
    if (wl != 0 || j >= HB_NPART) return;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][jl];
    if (j < HB_W * HB_W) dw1[j] = t;
    else if (j < HB_W * HB_W + HB_W) { if (db1) db1[j - HB_W * HB_W] = t; }
    else if (j < HB_W * HB_W + 2 * HB_W) dw2[j - HB_W * HB_W - HB_W] = t;
    else if (db2) db2[0] = t;
}

static int hb_grid(int64_t rows) {
    const int64_t nt = gml_cdiv(rows, HB_ROWS);
    return (int)(nt < GML_NUM_CU ? nt : GML_NUM_CU);
}

static int hb_check(const float* p, int64_t ldp, const float* y, const float* w1, const float* w2, int64_t rows, int64_t rows_loss,
                    int32_t nin, int32_t nh) {
    if (rows <= 0 || rows_loss < 0 || rows_loss > rows || nin <= 0 || nh <= 0 || ldp < nin) return GML_E_BADARG;
    if (nin != HB_W || nh != HB_W || ldp % 4 || ((uintptr_t)p & 15) || ((uintptr_t)w1 & 15)) return GML_E_UNSUPPORTED;
    if (!p || !y || !w1 || !w2) return GML_E_BADARG;
    return GML_OK;
}

// floats of workspace gml_head_l1_big_fwd / _bwd need (0: shape not served -- nin = nh = 32 only)
extern "C" size_t gml_head_l1_big_workspace_floats(int64_t rows, int32_t nin, int32_t nh) {
    if (rows <= 0 || nin != HB_W || nh != HB_W) return 0;
    return (size_t)hb_grid(rows) * HB_NPART;
}

// loss[0] = sum_{r < rows_loss} valid[r] |fc2(relu(fc1(p[r]))) - y[r]| for any number of rows (nin = nh = 32); loss_sum[0] += loss (optional)
extern "C" int gml_head_l1_big_fwd(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                                   const float* w2, const float* b2, int64_t rows, int64_t rows_loss, int32_t nin, int32_t nh,
                                   float* loss, float* loss_sum, void* ws, size_t ws_floats, gml_stream_t stream) {
    const int rc = hb_check(p, ldp, y, w1, w2, rows, rows_loss, nin, nh);
    if (rc != GML_OK) return rc;
    if (!loss) return GML_E_BADARG;
    const int grid = hb_grid(rows);
    if (!ws || ws_floats < (size_t)grid) return GML_E_WORKSPACE;
    GmlHeadBigParams q = {};
    q.p = p; q.ldp = ldp; q.y = y; q.valid = valid; q.w1 = w1; q.b1 = b1; q.w2 = w2; q.b2 = b2; q.R = rows; q.Rl = rows_loss;
    q.ntiles = (int)gml_cdiv(rows, HB_ROWS); q.part = (float*)ws;
    hipLaunchKernelGGL(gml_k_headbig_fwd, dim3(grid), dim3(HB_ROWS), 0, (hipStream_t)stream, q);
    int st = gml_launch_status();
    if (st != GML_OK) return st;
    hipLaunchKernelGGL(gml_k_headbig_loss_fold, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, grid, loss, loss_sum);
    return gml_launch_status();
}

// every gradient of that loss times gscale[0] (NULL: 1): gp [rows, 32] (rows >= rows_loss receive zeros), dw1 [32, 32], db1 [32] (may
// be NULL), dw2 [32], db2 [1] (may be NULL).  The forward is recomputed from p.  dw1 = dw2 = NULL: the per-workgroup partials
// [parts][32 * 32 + 32 + 32 + 1] (dw1 | db1 | dw2 | db2; parts = gml_head_l1_big_workspace_floats / 1089) stay in ws for gml_fold_many.
extern "C" int gml_head_l1_big_bwd(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                                   const float* w2, const float* b2, int64_t rows, int64_t rows_loss, int32_t nin, int32_t nh,
                                   const float* gscale, float* gp, int64_t ldgp, float* dw1, float* db1, float* dw2, float* db2,
                                   void* ws, size_t ws_floats, gml_stream_t stream) {
    const int rc = hb_check(p, ldp, y, w1, w2, rows, rows_loss, nin, nh);
    if (rc != GML_OK) return rc;
    if (!gp || ldgp < nin || (dw1 == nullptr) != (dw2 == nullptr)) return GML_E_BADARG;
    if (ldgp % 4 || ((uintptr_t)gp & 15)) return GML_E_UNSUPPORTED;
    const int grid = hb_grid(rows);
    if (!ws || ws_floats < (size_t)grid * HB_NPART) return GML_E_WORKSPACE;
    GmlHeadBigParams q = {};
    q.p = p; q.ldp = ldp; q.y = y; q.valid = valid; q.w1 = w1; q.b1 = b1; q.w2 = w2; q.b2 = b2; q.R = rows; q.Rl = rows_loss;
    q.ntiles = (int)gml_cdiv(rows, HB_ROWS); q.part = (float*)ws; q.gscale = gscale; q.gp = gp; q.ldgp = ldgp;
    const size_t lds = sizeof(float) * ((size_t)HB_W * HB_W + 2 * HB_W + (size_t)HB_ROWS * HB_LDA + (size_t)HB_ROWS * HB_LDP);
    GML_ALLOW_BIG_LDS(rca, (&gml_k_headbig_bwd), 160 * 1024)
    if (rca != hipSuccess) return (int)rca;
    hipLaunchKernelGGL(gml_k_headbig_bwd, dim3(grid), dim3(HB_ROWS), lds, (hipStream_t)stream, q);
    int st = gml_launch_status();
    if (st != GML_OK || !dw1) return st;
    hipLaunchKernelGGL(gml_k_headbig_fold, dim3((unsigned)gml_cdiv(HB_NPART, 16)), dim3(256), 0, (hipStream_t)stream, (const float*)ws,
                       (int64_t)grid, dw1, db1, dw2, db2);
    return gml_launch_status();
}
