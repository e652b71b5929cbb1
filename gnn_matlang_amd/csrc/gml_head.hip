// Readout head + L1-sum loss of the ZINC GNNML3 in two launches (reference: /root/reference/Zinc12k.py:343-345 head, :365 loss)
//
//   h = relu(W1 p + b1),   pre = w2 . h + b2,   loss = sum_r valid[r] |pre_r - y_r|          p = pooled features [R, nin]
//
// for the reference's regime -- a batch of 64 graphs: 65 pooled rows -- where the step is a chain of launches and the head, the loss
// and their backward were ~20 of its 70 (two library GEMMs + bias + relu + sub / abs / mask / sum forward; sign, mask, three
// GEMMs, two column sums, a relu mask backward; profiles/r04_epoch_bs64_kernel_trace.md).  One workgroup; everything in LDS; fixed
// summation orders (bitwise repeatable).  Rows beyond nvalid_rows (the padding graph of a static batch) take no part in the
// loss and receive a zero gradient.
#include "gml_common.h"

#define GML_HEAD_MAX_ROWS 256
#define GML_HEAD_MAX_W 64

struct GmlHeadParams {
    const float* p; int64_t ldp;
    const float* y; const float* valid;                      // [nrows_loss]; valid may be NULL (all ones)
    const float* w1; const float* b1; const float* w2; const float* b2;   // [nh, nin], [nh], [1, nh], [1]
    int32_t R, Rl, nin, nh;                                    // pooled rows, rows that enter the loss (<= R)
    float* loss;                                               // forward: scalar out
    float* loss_sum;                                           // forward: optional running sum (loss_sum[0] += loss: an epoch's loss without a launch of its own)
    float* pre;                                                // forward: [R] logits out (may be NULL)
    const float* gscale;                                       // backward: upstream gradient of the loss (device scalar; NULL = 1)
    float* gp; int64_t ldgp;                                   // backward: d loss / d p [R, nin]
    float* dw1; float* db1; float* dw2; float* db2;
};

__device__ __forceinline__ float gml_dot4(const f32x4 a, const f32x4 b, float acc) {
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); return fmaf(a.w, b.w, acc);
}

// hidden activations and logits of every row into LDS: hs[r][j] = relu(.), pre_l[r].  nin, nh are multiples of 4 (checked by the host):
// every LDS access is a 16-byte one and the short loops are unrolled, so the reads of a dot product are in flight together (the
// first version walked them one dependent 4-byte read at a time: 36 us for 65 rows)
__device__ __forceinline__ void gml_head_forward(const GmlHeadParams& q, float* ps, float* hs, float* pre_l, const float* w1s) {
    const int tid = threadIdx.x, NT = blockDim.x;
    const int n4 = q.nin / 4, h4 = q.nh / 4;
    for (int i = tid; i < q.R * n4; i += NT)
        *reinterpret_cast<f32x4*>(ps + 4 * i) = *reinterpret_cast<const f32x4*>(q.p + (int64_t)(i / n4) * q.ldp + 4 * (i % n4));
    __syncthreads();
    for (int i = tid; i < q.R * q.nh; i += NT) {
        const int r = i / q.nh, j = i % q.nh;
        float a = q.b1 ? q.b1[j] : 0.f;
#pragma unroll 8
        for (int k = 0; k < n4; ++k)
            a = gml_dot4(*reinterpret_cast<const f32x4*>(w1s + j * q.nin + 4 * k), *reinterpret_cast<const f32x4*>(ps + r * q.nin + 4 * k), a);
        hs[i] = fmaxf(a, 0.f);
    }
    __syncthreads();
    for (int r = tid; r < q.R; r += NT) {
        float a = q.b2 ? q.b2[0] : 0.f;
#pragma unroll 8
        for (int j = 0; j < h4; ++j)
            a = gml_dot4(*reinterpret_cast<const f32x4*>(q.w2 + 4 * j), *reinterpret_cast<const f32x4*>(hs + r * q.nh + 4 * j), a);
        pre_l[r] = a;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void gml_k_head_l1_fwd(const GmlHeadParams q) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int R4 = (q.R + 3) & ~3;                             // (every region 16-byte aligned)
    float* ps = sm; float* hs = ps + q.R * q.nin; float* pre_l = hs + q.R * q.nh; float* w1s = pre_l + R4; float* red = w1s + q.nh * q.nin;
    for (int i = threadIdx.x; i < q.nh * q.nin / 4; i += blockDim.x) *reinterpret_cast<f32x4*>(w1s + 4 * i) = *reinterpret_cast<const f32x4*>(q.w1 + 4 * i);
    gml_head_forward(q, ps, hs, pre_l, w1s);
    if (q.pre) for (int r = threadIdx.x; r < q.R; r += blockDim.x) q.pre[r] = pre_l[r];
    // loss: fixed-order tree over 256 partial sums
    float a = 0.f;
    for (int r = threadIdx.x; r < q.Rl; r += blockDim.x) a += fabsf(pre_l[r] - q.y[r]) * (q.valid ? q.valid[r] : 1.f);
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        q.loss[0] = red[0];
        if (q.loss_sum) q.loss_sum[0] += red[0];
    }
}

__global__ __launch_bounds__(256) void gml_k_head_l1_bwd(const GmlHeadParams q) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int R4 = (q.R + 3) & ~3;
    float* ps = sm; float* hs = ps + q.R * q.nin; float* pre_l = hs + q.R * q.nh; float* w1s = pre_l + R4;
    float* dh = w1s + q.nh * q.nin;                           // [R][nh]
    float* sg = dh + q.R * q.nh;                              // [R]
    const int tid = threadIdx.x, NT = blockDim.x;
    for (int i = tid; i < q.nh * q.nin / 4; i += NT) *reinterpret_cast<f32x4*>(w1s + 4 * i) = *reinterpret_cast<const f32x4*>(q.w1 + 4 * i);
    gml_head_forward(q, ps, hs, pre_l, w1s);
    const float gs = q.gscale ? q.gscale[0] : 1.f;
    for (int r = tid; r < q.R; r += NT) {
        float s = 0.f;
        if (r < q.Rl) {
            const float d = pre_l[r] - q.y[r];
            s = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (q.valid ? q.valid[r] : 1.f) * gs;     // torch.sign: 0 at 0
        }
        sg[r] = s;
    }
    __syncthreads();
    for (int i = tid; i < q.R * q.nh; i += NT) {
        const int r = i / q.nh, j = i % q.nh;
        dh[i] = hs[i] > 0.f ? sg[r] * q.w2[j] : 0.f;
    }
    __syncthreads();
    const int n4 = q.nin / 4;
    // d loss / d p [R, nin]: thread <-> (row, 4 consecutive inputs)
    for (int i = tid; i < q.R * n4; i += NT) {
        const int r = i / n4, k = 4 * (i % n4);
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int j = 0; j < q.nh; ++j) a += dh[r * q.nh + j] * *reinterpret_cast<const f32x4*>(w1s + j * q.nin + k);
        *reinterpret_cast<f32x4*>(q.gp + (int64_t)r * q.ldgp + k) = a;
    }
    // dW1[j][k] = sum_r dh[r][j] p[r][k]  (thread <-> (unit, 4 consecutive inputs));  db1[j] = sum_r dh[r][j];
    // dw2[j] = sum_r sg[r] h[r][j];  db2 = sum_r sg[r]   -- all in ascending r
    for (int i = tid; i < q.nh * n4; i += NT) {
        const int j = i / n4, k = 4 * (i % n4);
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int r = 0; r < q.R; ++r) a += dh[r * q.nh + j] * *reinterpret_cast<const f32x4*>(ps + r * q.nin + k);
        *reinterpret_cast<f32x4*>(q.dw1 + j * q.nin + k) = a;
    }
    for (int j = tid; j < q.nh; j += NT) {
        float a = 0.f, b = 0.f;
#pragma unroll 8
        for (int r = 0; r < q.R; ++r) { a += dh[r * q.nh + j]; b = fmaf(sg[r], hs[r * q.nh + j], b); }
        if (q.db1) q.db1[j] = a;
        q.dw2[j] = b;
    }
    if (tid == NT - 1 && q.db2) {
        float a = 0.f;
        for (int r = 0; r < q.R; ++r) a += sg[r];
        q.db2[0] = a;
    }
}

static int head_check(const GmlHeadParams& q) {
    if (q.R <= 0 || q.Rl < 0 || q.Rl > q.R || q.nin <= 0 || q.nh <= 0 || q.ldp < q.nin) return GML_E_BADARG;
    if (q.R > GML_HEAD_MAX_ROWS || q.nin > GML_HEAD_MAX_W || q.nh > GML_HEAD_MAX_W || q.nin % 4 || q.nh % 4 || q.ldp % 4 ||
        ((uintptr_t)q.p & 15) || ((uintptr_t)q.w1 & 15) || ((uintptr_t)q.w2 & 15)) return GML_E_UNSUPPORTED;
    if (!q.p || !q.y || !q.w1 || !q.w2) return GML_E_BADARG;
    return GML_OK;
}

// loss[0] = sum_{r < rows_loss} valid[r] |fc2(relu(fc1(p[r]))) - y[r]|; pre (optional) receives the logits of all `rows` rows.
// One workgroup: rows <= 256, nin, nh <= 64 (GML_E_UNSUPPORTED beyond: the caller uses its general path).
static int head_l1_fwd_impl(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                            const float* w2, const float* b2, int32_t rows, int32_t rows_loss, int32_t nin, int32_t nh,
                            float* loss, float* loss_sum, float* pre, gml_stream_t stream) {
    GmlHeadParams q = {};
    q.p = p; q.ldp = ldp; q.y = y; q.valid = valid; q.w1 = w1; q.b1 = b1; q.w2 = w2; q.b2 = b2;
    q.R = rows; q.Rl = rows_loss; q.nin = nin; q.nh = nh; q.loss = loss; q.loss_sum = loss_sum; q.pre = pre;
    const int rc = head_check(q);
    if (rc != GML_OK) return rc;
    if (!loss) return GML_E_BADARG;
    const size_t lds = sizeof(float) * ((size_t)rows * nin + (size_t)rows * nh + rows + 4 + (size_t)nh * nin + 256);
    if (lds > 160 * 1024) return GML_E_UNSUPPORTED;
    GML_ALLOW_BIG_LDS(rca, (&gml_k_head_l1_fwd), 160 * 1024)
    if (rca != hipSuccess) return (int)rca;
    hipLaunchKernelGGL(gml_k_head_l1_fwd, dim3(1), dim3(256), lds, (hipStream_t)stream, q);
    return gml_launch_status();
}

extern "C" int gml_head_l1_fwd(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                               const float* w2, const float* b2, int32_t rows, int32_t rows_loss, int32_t nin, int32_t nh,
                               float* loss, float* pre, gml_stream_t stream) {
    return head_l1_fwd_impl(p, ldp, y, valid, w1, b1, w2, b2, rows, rows_loss, nin, nh, loss, nullptr, pre, stream);
}

// the same + loss_sum[0] += loss (a running epoch loss -- Zinc12k.py:366 accumulates loss.item() on the host -- without a launch of its own)
extern "C" int gml_head_l1_fwd_acc(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                                   const float* w2, const float* b2, int32_t rows, int32_t rows_loss, int32_t nin, int32_t nh,
                                   float* loss, float* loss_sum, float* pre, gml_stream_t stream) {
    return head_l1_fwd_impl(p, ldp, y, valid, w1, b1, w2, b2, rows, rows_loss, nin, nh, loss, loss_sum, pre, stream);
}

// every gradient of the loss above times gscale[0] (NULL: 1): gp [rows, nin] (rows >= rows_loss receive zeros), dw1 [nh, nin], db1 [nh]
// (may be NULL), dw2 [nh], db2 [1] (may be NULL).  Recomputes the forward from p (nothing was saved).
extern "C" int gml_head_l1_bwd(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                               const float* w2, const float* b2, int32_t rows, int32_t rows_loss, int32_t nin, int32_t nh,
                               const float* gscale, float* gp, int64_t ldgp, float* dw1, float* db1, float* dw2, float* db2,
                               gml_stream_t stream) {
    GmlHeadParams q = {};
    q.p = p; q.ldp = ldp; q.y = y; q.valid = valid; q.w1 = w1; q.b1 = b1; q.w2 = w2; q.b2 = b2;
    q.R = rows; q.Rl = rows_loss; q.nin = nin; q.nh = nh; q.gscale = gscale;
    q.gp = gp; q.ldgp = ldgp; q.dw1 = dw1; q.db1 = db1; q.dw2 = dw2; q.db2 = db2;
    const int rc = head_check(q);
    if (rc != GML_OK) return rc;
    if (!gp || !dw1 || !dw2 || ldgp < nin) return GML_E_BADARG;
    if (ldgp % 4 || ((uintptr_t)gp & 15) || ((uintptr_t)dw1 & 15)) return GML_E_UNSUPPORTED;
    const size_t lds = sizeof(float) * ((size_t)rows * nin + 2 * (size_t)rows * nh + 2 * (size_t)rows + 8 + (size_t)nh * nin);
    if (lds > 160 * 1024) return GML_E_UNSUPPORTED;
    GML_ALLOW_BIG_LDS(rca, (&gml_k_head_l1_bwd), 160 * 1024)
    if (rca != hipSuccess) return (int)rca;
    hipLaunchKernelGGL(gml_k_head_l1_bwd, dim3(1), dim3(256), lds, (hipStream_t)stream, q);
    return gml_launch_status();
}
