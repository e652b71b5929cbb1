/*
 * gml.h -- C ABI of libgml_hip.so: the MI355X (gfx950) kernels behind the GNNML1/GNNML3
 * spectral layer (SpectConv / ML3Layer of balcilar/gnn-matlang).
 *
 * Conventions
 *   - every entry point returns int: 0 = ok, >0 = hipError_t, <0 = GML_E_* argument error;
 *   - no allocation, no global state, no synchronisation inside: the caller owns every buffer
 *     (device pointers), passes the hipStream_t the work is ordered on, and sizes scratch with
 *     the *_workspace_bytes helpers;
 *   - all matrices are row-major fp32 with an explicit leading dimension in ELEMENTS;
 *     node ids / edge positions are int32 on the device (int64 only at the COO boundary, which
 *     is what the reference hands over);
 *   - inputs are never written.
 *
 * What each entry replaces in the reference (pure Python, so the "FFI" a maintainer binds is
 * ctypes -- see INTEGRATION.md):
 *   gml_csr_*            nothing (the reference hands edge_index [2,E] int64 straight to PyG
 *                        MessagePassing.propagate, libs/spect_conv.py:77); the CSR keeps PyG's
 *                        per-target summation order (stable in input-edge order)
 *   gml_spectconv_fwd    SpectConv.forward default branch, libs/spect_conv.py:68-80,93-96 +
 *                        message :98-99 + PyG propagate (gather, scale, scatter-add) + matmul
 *   gml_spmm_fwd         the S propagate() calls alone (libs/spect_conv.py:77): H_s = A_s^T X
 *   gml_sddmm            autograd of message() w.r.t. norm (libs/spect_conv.py:98-99)
 *   gml_edge_mlp_*       ML3Layer.forward edge branch, libs/spect_conv.py:205-207
 *   gml_node_mix_*       ML3Layer.forward Hadamard branch, libs/spect_conv.py:209 (tanh*tanh)
 *   gml_relu_bwd / gml_segment_sum   glue the fused layer needs around the above
 */
#ifndef GML_H_
#define GML_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gml_stream_t; /* hipStream_t */

enum {
    GML_OK = 0,
    GML_E_BADARG = -1,       /* null pointer / negative size / bad stride */
    GML_E_UNSUPPORTED = -2,  /* shape outside the compiled kernel set */
    GML_E_WORKSPACE = -3     /* workspace too small */
};

/* flags of gml_spectconv_fwd */
enum {
    GML_RELU = 1,   /* out = max(out, 0) in the epilogue          (ML3Layer: relu(conv1(..))) */
    GML_ACCUM = 2,  /* out += result instead of out = result      (gradient accumulation)     */
    GML_F32_MFMA = 4, /* project with the f32-input MFMA (bit-identical to an fmaf chain) instead of the default
                         bf16x3 split on the bf16 matrix cores (fp32-class: ~1e-6 of the output scale)       */
    GML_GROUPS128 = 8, /* gml_spectconv_fwd: `ginfo` holds 128-row group records (gml_spectconv_fwd_group_rows() = 128):
                         the 8-wave forward kernel                                                            */
    GML_GROUPS64R = 16, /* gml_spectconv_fwd: `ginfo` holds ranked 64-row records (gml_spectconv_fwd_group_rows() =
                         GML_GROUPS64_RANKED): the same kernel in its 4-wave geometry, two workgroups per CU     */
    GML_FWD_CHUNKED = 64, /* gml_spectconv_fwd / gml_ml3_fwd with GML_GROUPS128: some 128-row group holds more edges than the ring
                         kernel stages at once (the caller knows the maximum from its group records): take the chunked ring kernel
                         (gml_k_spectconv_fwd4), which walks such groups in edge chunks instead of gathering from global memory */
    GML_DVAL_ACCUM = 128, /* gml_spectconv_bwd (8-wave bf16x3 kernel): dval += instead of dval = -- the second of two launches over
                         slices of the input features (48-wide layers: features 0..31, then 32..47; dval is linear in x) */
    GML_FWD_ONEWIN = 256, /* gml_spectconv_fwd, 48-feature shapes on the chunked ring kernel: ONE staged X window instead of two, twice the
                         edges per work item -- for batches whose groups need edge chunks (the caller knows the batch's largest group) */
    GML_DMA_RING = 32,  /* gml_spectconv_bwd / _bwd_mix: take the LDS-DMA landing-ring kernel (bwd4) where it applies; the
                         forward uses its ring kernel (fwd3) by default (GML_FWD_DMA=0 in the environment turns it off)   */
    GML_F16X3 = 1024,   /* gml_spectconv_fwd / gml_ml3_fwd on the 8-wave kernels (fwd3, its register-staged form fwd2, the chunked ring kernel fwd4): project with f16 (hi, lo) pieces under per-tile /
                         per-column power-of-two scales instead of bf16 pairs: residual 2^-24 instead of 2^-17 per operand, same
                         instruction count on the matrix pipe (csrc/gml_common.h "f16x3").  Ignored by the 64-row kernel family. */
    GML_NO_FOLD = 512   /* gml_spectconv_bwd / _bwd_mix / _bwd_mix_relu with dw != NULL: dw is NOT written -- the per-workgroup partial
                         sums stay in ws as [parts][S * Fin * Fout], parts = gml_spectconv_bwd_workspace_bytes(...) / (4 S Fin Fout) --
                         for a later gml_fold_many (the reference's batch 64: twelve fold launches of a step become one) */
};

/* gml_spectconv_bwd_mix_relu with the rows of wmix in two arrays (rows [0, nmix_a) from wmix_a, the next nmix_b from wmix_b): an
 * ML3Layer's fc11.weight and fc12.weight as they are stored -- no concatenation launch per layer and step. */
int gml_spectconv_bwd_mix_relu2(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                                const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                                float* dx, int64_t lddx, float* dval, float* dw, const float* dz, const float* wmix_a,
                                int32_t nmix_a, const float* wmix_b, int32_t nmix_b, int32_t relu_cols, int64_t num_rows,
                                int32_t S, int32_t Fin, int32_t Fout, int32_t max_group_edges, int32_t max_group_window,
                                uint32_t flags, void* ws, size_t ws_bytes, gml_stream_t stream);

/* gml_spectconv_bwd_mix_relu2 with the whole output stage of the ML3Layer inside -- the backward of
 *   out = cat[ relu(conv(x)) , tanh(fc11 x) * tanh(fc12 x) ]          (libs/spect_conv.py:209-212)
 * for a layer whose output gradient arrives pre-masked (the consumer layer's conv backward applied this layer's relu mask,
 * relu_cols above): ONE launch instead of gml_ml3_split_bwd_ex + gml_spectconv_bwd_mix_relu2.  g [num_rows, ldg >= Fout + F2] is
 * that gradient (columns [0, Fout) at the conv output, columns Fout, Fout + 1 at the Hadamard units); neither a dz array nor a second
 * pass over g and x exists -- dz is recomputed per 128-row group from the x rows the kernel holds, dw11 / dw12 ride in the kernel's own
 * row contraction (bf16x3 products like dw), the bias gradients are folded per wave in fixed order.  Shape class: S = 8,
 * 17 .. 32 input features (multiple of 4), Fout = 30, F2 = 2 (Zinc12k.py:338-341); gml_spectconv_bwd_had_parts returns 0 outside it
 * (and with GML_BWD_HAD=0 in the environment), otherwise the number of partial rows [dw11 | dw12 | db11 | db12 | dcb]
 * (4 Fin + 4 + Fout floats each) hws must hold.  dcb = dw11 = db11 = dw12 = db12 = NULL: the partials stay in hws for gml_fold_many;
 * otherwise they are folded into the given ones (dw11, dw12 required then; b11 / db11, b12 / db12 may be NULL).  dw required.
 * dx = NULL (want_dx = 0; the model's first layer, whose input is data): no dX is formed; then 17 .. 32 input features of any count,
 * x rows float4-readable up to roundup4(Fin) (ldx % 4 == 0, 16-byte aligned base), relu_cols = 0. */
int gml_spectconv_bwd_had_parts(int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout, int32_t F2, int32_t want_dx,
                                int32_t max_group_edges, int32_t max_group_window, uint32_t flags);
int gml_spectconv_bwd_had(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                          const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                          float* dx, int64_t lddx, float* dval, float* dw, const float* w11, const float* b11,
                          const float* w12, const float* b12, int32_t relu_cols, float* dcb, float* dw11, float* db11,
                          float* dw12, float* db12, int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout, int32_t F2,
                          int32_t max_group_edges, int32_t max_group_window, uint32_t flags, void* ws, size_t ws_bytes,
                          void* hws, size_t hws_bytes, gml_stream_t stream);

/* Deferred folds.  Every weight-gradient kernel of the library leaves one partial sum per workgroup in its workspace and folds
 * them in a second launch.  At the reference's batch size those folds are launches of a few microseconds of work each; a caller
 * may skip them -- GML_NO_FOLD above; gml_ml3_split_bwd(_ex) with dcb = dw11 = db11 = dw12 = db12 = NULL (partials [parts][2 F2 Fin +
 * 2 F2 + nout1] in the order dw11, dw12, db11, db12, dcb; parts = gml_ml3_split_bwd_workspace_bytes / row bytes); gml_edge_mlp_bwd
 * with dw1 = dw2 = dw3 = dw4 = NULL (partials [gml_edge_mlp_bwd_parts(...)][6 S^2 + 4 S Sout] in the order dw1..dw4) -- and fold all
 * of them with ONE launch:  dst[k][i] = sum_{w < nparts} partial[w * n + off_k + i]  in ascending w (the order of the per-kernel
 * folds: bit-identical results), off_k = ndst[0] + .. + ndst[k-1]; a NULL dst[k] skips its segment. */
typedef struct gml_fold_job {
    const float* partial; int64_t nparts; int64_t n;
    float* dst[5]; int64_t ndst[5];
} gml_fold_job;
#define GML_FOLD_MAX_JOBS 16
int gml_fold_many(const gml_fold_job* jobs /* host array */, int32_t njobs /* <= GML_FOLD_MAX_JOBS */, gml_stream_t stream);
/* Adam (torch.optim.Adam's update as the reference scripts configure it, Zinc12k.py:349: no weight decay, no amsgrad) over a list of
 * tensors in ONE launch: t = step[0] + 1; m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p -= lr / (1 - b1^t) m / (sqrt(v / (1 - b2^t)) + eps);
 * step[0] = t afterwards.  step [1] fp32 and done [1] uint32 (zero) live on the device: the launch is capturable, every replay of a
 * captured step advances the count.  More than GML_ADAM_MAX_JOBS tensors: several calls -- all but the last with a scratch copy of
 * step (the count must advance once per optimizer step). */
typedef struct gml_adam_job { float* p; const float* g; float* m; float* v; int64_t n; } gml_adam_job;
#define GML_ADAM_MAX_JOBS 64
int gml_adam_many(const gml_adam_job* jobs /* host array */, int32_t njobs, float* step, uint32_t* done, float lr, float beta1,
                  float beta2, float eps, gml_stream_t stream);
/* number of partial rows gml_edge_mlp_bwd leaves for this call shape (it depends on the kernel family the call takes) */
int64_t gml_edge_mlp_bwd_parts(int64_t num_edges, int32_t S, int32_t Sout, int32_t has_split, int32_t want_gin);

int gml_version(void);
/* static string for any return code of this library (hipGetErrorString for >0) */
const char* gml_error_string(int code);

/* ---------------------------------------------------------------- CSR build (integer-exact)
 * Stable sort of the COO edge list by `key` (key = dst for the forward view, key = src for the
 * transposed view): rowptr[N+1], other[E] = the non-key endpoint of each sorted edge, perm[E] =
 * input edge id of each sorted edge (ascending inside a row).  ws: gml_csr_workspace_bytes().
 * Node ids outside [0, num_nodes) are clamped (no out-of-bounds access) and reported: the LAST int32 of the workspace
 * (byte offset gml_csr_workspace_bytes() - 4) is OR-ed with 1.  The caller zeroes that word before the call(s) that share
 * the workspace and raises when it reads non-zero (the reference's scatter raises an index error for such input). */
size_t gml_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges);
int gml_csr_from_coo(const int64_t* key, const int64_t* other_in, int64_t num_nodes, int64_t num_edges,
                     int32_t* rowptr, int32_t* other, int32_t* perm,
                     void* ws, size_t ws_bytes, gml_stream_t stream);
/* The same for keys that are ALREADY non-decreasing (edge_index2 as the reference's transform emits it is sorted by
 * source, libs/utils.py:608-609): no sort -- rowptr from the run boundaries, perm = identity, one pass.  If a key is
 * smaller than its predecessor, bit 1 (value 2) of the workspace's flag word is set and the outputs are unspecified: the
 * caller must then use gml_csr_from_coo for this view. */
int gml_csr_from_sorted_coo(const int64_t* key, const int64_t* other_in, int64_t num_nodes, int64_t num_edges,
                            int32_t* rowptr, int32_t* other, int32_t* perm,
                            void* ws, size_t ws_bytes, gml_stream_t stream);
/* pos_t[j] = inverse(perm_fwd)[perm_t[j]]: where, in forward (target-sorted) order, the j-th
 * source-sorted edge keeps its values.  inv_scratch: E int32. */
int gml_csr_link_transpose(const int32_t* perm_fwd, const int32_t* perm_t, int64_t num_edges,
                           int32_t* inv_scratch, int32_t* pos_t, gml_stream_t stream);
/* Group records: the staging schedule of the fused kernels (gml_spectconv_fwd: group_rows = 64;
 * gml_spectconv_bwd: gml_spectconv_bwd_group_rows()).  One record of gml_csr_group_record_ints(group_rows) int32
 * per group of `group_rows` (64 or 128) consecutive rows, ceil(num_rows / group_rows) records:
 *   {first edge, #edges, smallest column id, column-window width},
 *   128-row groups: then 128 bytes, the local row each lane position of the backward kernel works on -- rows ranked
 *   by degree so that the 16 rows of a tile run near-equal edge loops, rank blocks dealt to the waves so the SIMDs
 *   stay balanced (which lane serves a row never changes the row's result);
 *   group_rows = GML_GROUPS64_RANKED (1064): 64-row groups that carry the same rank bytes (64 of them) -- the staging
 *   schedule of the 4-wave backward kernel that runs two workgroups per CU.
 * Callers size LDS with the maxima of ints 1 and 3 over the records.  gml_csr_group_record_ints = int32 per record. */
#define GML_GROUPS64_RANKED 1064
int32_t gml_csr_group_record_ints(int32_t group_rows);
int gml_csr_group_info(const int32_t* rowptr, const int32_t* col, int64_t num_rows, int32_t group_rows,
                       int32_t* ginfo, gml_stream_t stream);
/* the same for two CSR views over the same rows (target- and source-keyed) in ONE launch */
int gml_csr_group_info2(const int32_t* rowptr_a, const int32_t* col_a, int32_t* ginfo_a, const int32_t* rowptr_b,
                        const int32_t* col_b, int32_t* ginfo_b, int64_t num_rows, int32_t group_rows, gml_stream_t stream);
/* One launch: a padded static-shape batch (the graphs `ids`, in that order) and its whole index structure from a device-resident data
 * set whose per-graph structure was computed once.  The reference collates every batch on the host (DataLoader, Zinc12k.py:20-22,359);
 * here a batch is gathers + offsets, because a batch is the block-diagonal union of graphs whose own structure never changes.
 * Data set side (per graph g its nodes node_ptr[g]..node_ptr[g+1] and support edges edge_ptr2[g]..edge_ptr2[g+1], graph-LOCAL node ids,
 * sorted by source inside a graph): tperm[e] = source-order position (inside its graph) of the graph's k-th TARGET-sorted edge (stable),
 * tinv its inverse, rp_src / rp_dst [node] = number of the graph's edges whose source / target is a smaller node (local row pointers),
 * es = gml_edge_presplit(edge_attr2) or NULL.  ids entries outside [0, G) = no graph.  Outputs: x_out [n_pad, F], ea_out [e2_pad, S],
 * es_out [e2_pad, 8], y_out [B + 1], valid_out [B], ptr_out [B + 2], batch_out [n_pad] (padding nodes: graph B), and the CSR arrays of
 * both views exactly as gml_csr_from_coo / _from_sorted_coo / _link_transpose give them for the assembled batch (perm == tpos;
 * perm_t is the identity).  Padding: zero-feature nodes forming graph B; zero-valued self loops dealt dmax per padding node. */
typedef struct gml_batch_desc {
    const int64_t* node_ptr; const int64_t* edge_ptr2; const float* x; const int64_t* edge_index2; const float* edge_attr2;
    const int32_t* es; const int32_t* tperm; const int32_t* tinv; const int32_t* rp_src; const int32_t* rp_dst; const float* y;
    int64_t G, E2all; int32_t F, S;
    const int64_t* ids; int32_t B, n_pad, e2_pad, dmax;
    float* x_out; float* ea_out; int32_t* es_out; float* y_out; float* valid_out; int32_t* ptr_out; int32_t* batch_out;
    int32_t* rowptr; int32_t* col; int32_t* perm; int32_t* rowptr_t; int32_t* col_t; int32_t* pos_t;
    int32_t ldx_out;   /* floats between rows of x_out (0: F); columns F .. ldx_out - 1 are written as zeros (float4-addressable rows) */
    int32_t nblk_main; /* (set by the library) */
    /* optional (both or neither): the 128-row group records of the target-keyed / source-keyed view, [ceil(n_pad / 128)][gml_csr_group_record_ints(128)]
       ints each -- what gml_csr_group_info2 would compute from the assembled index arrays, written by the same launch (round 5) */
    int32_t* ginfo128; int32_t* ginfo_t128;
} gml_batch_desc;
int gml_batch_assemble(const gml_batch_desc* d, gml_stream_t stream);
/* out[k, :] = in[perm[k], :]   (rows of `width` floats) */
int gml_gather_rows(const float* in, const int32_t* perm, float* out, int64_t rows, int32_t width,
                    gml_stream_t stream);
/* out[k, :] = in[perm[k], :] and, in the same pass, its bf16 pre-split (gml_edge_presplit of out); S <= 8 */
int gml_gather_rows_presplit(const float* in, const int32_t* perm, float* out, void* out_split, int64_t rows,
                             int32_t S, gml_stream_t stream);
/* out[perm[k], :] = in[k, :] */
int gml_scatter_rows(const float* in, const int32_t* perm, float* out, int64_t rows, int32_t width,
                     gml_stream_t stream);

/* ---------------------------------------------------------------- fused multi-support layer
 *   out[r, 0:Fout] (op)= act( sum_s ( sum_{k in row r} val[pos(k), s] * x[col[k], :] ) @ W[s] + bias )
 * rowptr/col: CSR keyed by the OUTPUT row (ginfo: gml_csr_group_info of it); pos(k) = epos ? epos[k] : k
 * (epos = NULL, i.e. values stored in the order of this CSR, is the fast path); val is [E, S] (S contiguous).
 * W element (s, i, o) lives at w[s*w_ss + i*w_si + o*w_so] so the transposed weights of the
 * backward pass need no copy.  bias may be NULL.  Used for: forward (CSR by target, x = X),
 * d/dX (CSR by source, x = dOut, W transposed view, epos = pos_t). */
/* group size (64 | 128) whose records the forward wants for this shape and arithmetic; 128 additionally needs
 * epos == NULL -- the caller then passes those records and GML_GROUPS128 */
int32_t gml_spectconv_fwd_group_rows(int32_t S, int32_t Fin, int32_t Fout, uint32_t flags);
/* edges of one 128-row group the ring kernel of this shape keeps in LDS at once (0: not applicable).  When the largest group of the
 * batch (int 1 of its records) exceeds it: shapes of the default ring kernel gather such groups from global memory, or -- with
 * GML_FWD_CHUNKED -- run on the chunked ring kernel, which walks them in edge chunks (sr25.py: 13 entries per row); the shapes only the
 * chunked kernel serves (6 supports, 33..48 input features) always chunk (with GML_FWD_ONEWIN at 33..48 features: twice the edges per
 * chunk).  gnn_matlang_amd.functional takes the chunked road whenever a batch needs it (GML_FWD_CHUNKS=0 in the environment: the 64-row
 * family instead); its run-to-run stability is covered by tests/stress (DESIGN s4.1c). */
int32_t gml_spectconv_fwd_stage_edges(int32_t S, int32_t Fin, int32_t Fout, uint32_t flags);
/* widest column window (int 3 of the 128-row group records) the kernel this shape (and these flags, e.g. GML_FWD_CHUNKED) would run on
 * serves; 0 = no bound.  A batch with a wider group must use the 64-row family (64-row records, no GML_GROUPS128) for this call:
 * the chunked ring kernel has no road for such groups and writes NaN into their rows instead of wrong numbers. */
int32_t gml_spectconv_fwd_stage_window(int32_t S, int32_t Fin, int32_t Fout, uint32_t flags);
int gml_spectconv_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const int32_t* epos,
                      const float* val, const float* x, int64_t ldx,
                      const float* w, int64_t w_ss, int64_t w_si, int64_t w_so,
                      const float* bias, float* out, int64_t ldo,
                      int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                      uint32_t flags, gml_stream_t stream);

/* The two other forward forms of the reference's operator family on the ring kernel's epilogues (one launch, projection flops
 * 1x): epilogue = 1, SpectConCatConv.forward (libs/spect_conv.py:137-158): out [N, (S + self_term) Fout], support s writes column
 * block s + self_term (+ that block's bias; block 0 of a selfconn layer = x W_last is the caller's GEMM); epilogue = 2,
 * SpectConv.forward depthwise branch (libs/spect_conv.py:81-91): w is ONE [Fin, Fout] matrix (w_ss ignored), ds
 * [S + self_term, Fin] the per-feature scales (row 0 = 1 + DSweight[0]; last row: scale of the self term when self_term):
 * out = (sum_s ds_s . H_s + ds_self . x) W + bias.  ginfo128: 128-row group records.  GML_E_UNSUPPORTED outside S in {4, 8},
 * Fin, Fout <= 32, float4-addressable x rows: the caller then maps onto gml_spectconv_fwd through transformed weights. */
int gml_spectconv_fwd_epi(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo128, const float* val,
                          const float* x, int64_t ldx, const float* w, int64_t w_ss, int64_t w_si, int64_t w_so,
                          const float* bias, float* out, int64_t ldo, int64_t num_rows, int32_t S, int32_t Fin,
                          int32_t Fout, uint32_t flags, int32_t epilogue, const float* ds, int32_t self_term,
                          gml_stream_t stream);

/* ---------------------------------------------------------------- fused backward of the layer above
 * CSR keyed by SOURCE row (rowptr/col = targets, ginfo of it), val [E, S] in that order.
 *   dx[r, :]   (op)= sum_s ( sum_{e out of r} val[e, s] g[col[e], :] ) @ W[s]^T          (NULL: skip)
 *   dval[e, s]  =  < x[r, :] @ W[s], g[col[e], :] >   for e out of r, same order as val    (NULL: skip)
 *   dw[s]       =  sum_r x[r, :]^T ( sum_{e out of r} val[e, s] g[col[e], :] )           (NULL: skip)
 * g = gradient at the layer output (after the relu mask), [N, Fout].  max_group_edges / max_group_window
 * = maxima of ginfo[:, 1] / ginfo[:, 3] (the caller reads them once per batch): they size the LDS
 * staging.  gml_spectconv_bwd_workspace_bytes returns 0 when the shape has no fused backward
 * (gml_spectconv_bwd then returns GML_E_UNSUPPORTED): the caller composes gml_spectconv_fwd on the
 * transposed view + gml_spmm_fwd + gml_sddmm instead.  flags: GML_ACCUM applies to dx; GML_F32_MFMA forces the
 * f32-input MFMA kernel (default: bf16x3 split on the bf16 matrix cores where the shape allows).
 * With 128-row groups (see below) g must be float4-addressable: 16-byte aligned, ldg % 4 == 0 and
 * ldg >= roundup4(Fout) with the padding columns zero (else GML_E_BADARG). */
/* group kind of the backward kernel this shape / flags selects: 128 (bf16x3 kernel, 8 waves), GML_GROUPS64_RANKED
 * (bf16x3 kernel, 4 waves, two workgroups per CU), 64 (f32-MFMA kernel) or 0 (no fused backward).  ginfo,
 * max_group_edges and max_group_window passed to the two functions below must be those of
 * gml_csr_group_info(..., group_rows = this value). */
int gml_spectconv_bwd_group_rows(int32_t S, int32_t Fin, int32_t Fout, uint32_t flags);
size_t gml_spectconv_bwd_workspace_bytes(int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                                         int32_t max_group_edges, int32_t max_group_window, uint32_t flags);
int gml_spectconv_bwd(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                      const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                      float* dx, int64_t lddx, float* dval, float* dw,
                      int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                      int32_t max_group_edges, int32_t max_group_window, uint32_t flags,
                      void* ws, size_t ws_bytes, gml_stream_t stream);

/* gml_spectconv_bwd with  dx = conv part + dz wmix : dz [num_rows, 4] (contiguous, 16-byte aligned, columns >= nmix ignored),
 * wmix [nmix <= 4, Fin].  The ML3Layer Hadamard branch (2 nout2 <= 4 columns: Zinc12k.py's 30+2 layers) hands its share of
 * dx over as its pre-activation gradients (gml_ml3_split_bwd_ex) instead of a written-then-re-read [N, Fin] array.
 * dx must be wanted and float4-addressable; GML_ACCUM is not combined with it.  gml_spectconv_bwd_mix_supported: 1 when
 * the shape has this form (the 8-wave bf16x3 kernel, S = 8, 16 < Fin <= 32), else the caller uses dx + GML_ACCUM. */
int gml_spectconv_bwd_mix_supported(int32_t S, int32_t Fin, int32_t Fout, int32_t nmix, uint32_t flags);
int gml_spectconv_bwd_mix(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                          const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                          float* dx, int64_t lddx, float* dval, float* dw, const float* dz, const float* wmix, int32_t nmix,
                          int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                          int32_t max_group_edges, int32_t max_group_window, uint32_t flags,
                          void* ws, size_t ws_bytes, gml_stream_t stream);

/* gml_spectconv_bwd_mix for a layer whose input x is itself an ML3Layer output [relu(conv) | Hadamard columns] with nothing
 * else consuming it (libs/spect_conv.py:209-212 stacked as in Zinc12k.py:338-341): dx[:, f] for f < relu_cols is written
 * multiplied by (x[:, f] > 0), i.e. dx[:, :relu_cols] IS the gradient at the conv output of the layer below.  That layer then
 * calls gml_ml3_split_bwd_ex in its pre-masked form (y = G = NULL) and passes dx itself as g to its own conv backward
 * (columns >= its Fout are ignored there).  relu_cols = 0: gml_spectconv_bwd_mix.  Same support test. */
int gml_spectconv_bwd_mix_relu(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const float* val,
                               const float* x, int64_t ldx, const float* g, int64_t ldg, const float* w,
                               float* dx, int64_t lddx, float* dval, float* dw, const float* dz, const float* wmix, int32_t nmix,
                               int32_t relu_cols, int64_t num_rows, int32_t S, int32_t Fin, int32_t Fout,
                               int32_t max_group_edges, int32_t max_group_window, uint32_t flags,
                               void* ws, size_t ws_bytes, gml_stream_t stream);

/* H[r, s, :] = sum_{k in row r} val[pos(k), s] * x[col[k], :]     H is [N, S, Fin] contiguous.
 * ginfo128 (optional): 128-row group records of this CSR -> the LDS-DMA ring kernel (gml_spmm3_impl.h): any S, any Fin % 4 == 0,
 * float4-addressable x rows, epos NULL, groups of up to ~2048 staged edges (larger ones gather from global memory, same
 * results); NULL or an unsupported layout: one-row-per-lane-group kernel for any shape. */
int gml_spmm_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo128, const int32_t* epos,
                 const float* val, const float* x, int64_t ldx, float* h,
                 int64_t num_rows, int32_t S, int32_t Fin, gml_stream_t stream);
/* the same with a hint: max_group_edges = the largest edge count of a 128-row group (max of ginfo128[:, 1]; < 0 = unknown).
 * Only the kernel choice depends on it (the register-staged 8-wave kernel while every group fits its staging, else the ring
 * kernel), never the result. */
int gml_spmm_fwd_ex(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo128, const int32_t* epos,
                    const float* val, const float* x, int64_t ldx, float* h,
                    int64_t num_rows, int32_t S, int32_t Fin, int32_t max_group_edges, gml_stream_t stream);

/* dval[pos(k), s] = < x[col[k], :], gw[r, s, :] >  for k in row r;   gw is [N, S, Fin] contiguous */
int gml_sddmm(const int32_t* rowptr, const int32_t* col, const int32_t* epos,
              const float* x, int64_t ldx, const float* gw, float* dval,
              int64_t num_rows, int32_t S, int32_t Fin, gml_stream_t stream);

/* ---------------------------------------------------------------- ML3Layer edge branch
 *   out = relu( W4 . [ relu(W1 . e) ; tanh(W2 . e) * tanh(W3 . e) ] )      per edge e in R^S
 * w1,w2,w3: [2S, S]; w4: [Sout, 4S]  (torch.nn.Linear layout, bias-free).  S = Sout <= 16.
 * If out_t != NULL the row of edge e is also written to out_t[tpos[e], :] (the same values in a second
 * edge order: the backward kernel walks the source-sorted order); on the matrix-core kernels (2 <= S <= 8) that
 * scatter uses 32-bit byte offsets: num_edges * S * 4 must stay below 2^32 - 256, else GML_E_UNSUPPORTED (the
 * caller then permutes `out` itself).
 * ea_split (optional, S <= 8): the rows of ea split once into bf16 hi[8] | lo[8] (32 bytes per edge, 16-byte aligned)
 * by gml_edge_presplit -- the raw supports are per-batch constants, so the matrix-core kernels load their first
 * operand ready-made instead of splitting it per layer and per step.  Must describe the same ea (same edge order). */
int gml_edge_presplit(const float* ea, void* ea_split, int64_t num_edges, int32_t S, gml_stream_t stream);
int gml_edge_mlp_fwd(const float* ea, const void* ea_split, const float* w1, const float* w2, const float* w3,
                     const float* w4, float* out, const int32_t* tpos, float* out_t,
                     int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream);

/* The same function for MORE than 16 supports, 16 < max(S, Sout) <= 48 (csrc/gml_edge_wide.hip; SURVEY s8(d)'s sr25 sweep at 24 and
 * 48 supports; reference: libs/spect_conv.py:190-194, 205-207): one launch, exact fp32 products, weights resident in LDS, no
 * intermediate in HBM.  ea [num_edges, S], out [num_edges, Sout]; rows 16-byte aligned where S / Sout are multiples of 4.
 * GML_E_UNSUPPORTED for max(S, Sout) <= 16 (gml_edge_mlp_fwd) or > 48.  Forward only: the host recomputes through the library
 * for the backward (no reference script trains more than 12 supports). */
int gml_edge_mlp_wide_fwd(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, float* out,
                          int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream);
/* Its backward, per-edge part (round 6; the autograd of libs/spect_conv.py:205-207): out = the forward's result, gout = dL/dout;
 * writes go [E, Sout] = gout * (out > 0), hid [E, 2 H2R] = relu(W1 e) | tanh(W2 e) tanh(W3 e) and gz [E, 3 H2R] = the gradients at
 * the three pre-activations, H2R = gml_edge_mlp_wide_bwd_h2r(S) = 2 S rounded up to a multiple of 4 (rows 16-byte aligned).  The
 * weight gradients are the tall contractions dW_m = gz[:, m H2R : m H2R + 2 S]^T ea and dW4[:, b 2S : (b+1) 2S] = (hid[:, b H2R : b H2R
 * + 2 S]^T go)^T (gml_xty_wide); the supports' own gradient sum_m gz_m W_m.  Exact fp32 products; same shape range as the forward. */
int32_t gml_edge_mlp_wide_bwd_h2r(int32_t S);
int gml_edge_mlp_wide_bwd(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, const float* out,
                          const float* gout, float* go, float* hid, float* gz, int64_t num_edges, int32_t S, int32_t Sout,
                          gml_stream_t stream);

/* The edge branches of a STACK of ML3Layers in one pass: every layer of Zinc12k.py:338-341 / counting.py:361-366 receives the
 * same raw supports (data.edge_attr2), so L launches of gml_edge_mlp_fwd read them L times.  out[l] [num_edges, Sout] =
 * the branch of layer l (weights w1[l] .. w4[l]) applied to the rows whose split image is ea_split (gml_edge_presplit), same
 * edge order.  The five pointer arrays (nlayers entries each) live on the HOST.  GML_E_UNSUPPORTED outside S = Sout in {4, 8},
 * 2 <= nlayers <= 4 (or under GML_EDGE_VALU=1): call gml_edge_mlp_fwd per layer -- same results either way. */
int gml_edge_mlp_fwd_stack(const void* ea_split, int32_t nlayers, const float* const* w1, const float* const* w2,
                           const float* const* w3, const float* const* w4, float* const* out,
                           int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream);
/* The edge-branch FORWARD with fp32-class products on the bf16 matrix cores ("bf16x6": every operand cut into three bf16 pieces, the
 * products down to 2^-24 kept; csrc/gml_edge_chain6_impl.h; reference: libs/spect_conv.py:205-207, fp32 throughout).  Reads the fp32
 * rows of ea itself (no pre-split image).  gml_edge_mlp_fwd6: one layer, 2 <= S = Sout <= 16, out / tpos / out_t as gml_edge_mlp_fwd.
 * gml_edge_mlp_fwd_stack6: the layers of a stack in one pass (host pointer arrays as gml_edge_mlp_fwd_stack), S = Sout in {4, 8},
 * 1 <= nlayers <= 4.  GML_E_UNSUPPORTED outside those shapes.  The default forward of the host side since round 6: the two-piece
 * chain's ~5e-7 error on the learned supports is what moved trained-state gradients beyond 1e-4 of their term sums (DESIGN s6). */
int gml_edge_mlp_fwd6(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, float* out,
                      const int32_t* tpos, float* out_t, int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream);
int gml_edge_mlp_fwd_stack6(const float* ea, int32_t nlayers, const float* const* w1, const float* const* w2,
                            const float* const* w3, const float* const* w4, float* const* out,
                            int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream);
/* The edge branch over a batch's UNIQUE support rows (csrc/gml_edge_chain_sym_impl.h).  SpectralDesign's supports sample symmetric
 * matrices (libs/utils.py:546-610), so edge (i, j) and its mirror (j, i) mostly carry bitwise the same row and the branch
 * (libs/spect_conv.py:205-207) gives both the same output.  gml_edge_sym_flags: per edge of the SOURCE-keyed view (rowptr_t, col_t of
 * gml_csr_from_coo; val_s [num_edges, S] in that order) flag = 2 (evaluate, and the mirror at position mirror[k] takes the same
 * row: src < dst, rows bitwise equal), 0 (covered by its mirror) or 1 (evaluate alone); mirror = -1 unless flag = 2.  The caller
 * compacts the edges with flag > 0 into uid / mir [num_unique] (int32).  gml_edge_mlp_fwd_stack6_sym: gml_edge_mlp_fwd_stack6 over
 * those entries, out[l][uid[u]] and out[l][mir[u]] written (every row of out is written exactly once when uid / mir come from the flags).
 * gml_edge_mlp_bwd_sym: gml_edge_mlp_bwd (no gin) with gout[uid[u]] + gout[mir[u]] as the entry's output gradient; partial rows in ws:
 * gml_edge_mlp_bwd_sym_parts(num_unique, S) (ws sized by gml_edge_mlp_bwd_workspace_bytes(num_edges, ..) is large enough); dw1 .. dw4
 * all NULL leaves the partials for gml_fold_many.  2 <= S = Sout <= 16 (layer stacks: S in {4, 8}; S > 8: ea_split = the 64-byte rows of gml_edge_presplit); GML_E_UNSUPPORTED otherwise.  Exact: no tolerance --
 * rows that differ in one bit are evaluated separately. */
int gml_edge_sym_flags(const int32_t* rowptr_t, const int32_t* col_t, const float* val_s, int64_t num_rows, int64_t num_edges,
                       int32_t S, int32_t* flag, int32_t* mirror, gml_stream_t stream);
int gml_edge_mlp_fwd_stack6_sym(const float* ea, const int32_t* uid, const int32_t* mir, int64_t num_unique, int32_t nlayers,
                                const float* const* w1, const float* const* w2, const float* const* w3, const float* const* w4,
                                float* const* out, int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream);
int64_t gml_edge_mlp_bwd_sym_parts(int64_t num_unique, int32_t S);
int gml_edge_mlp_bwd_sym(const void* ea_split, const int32_t* uid, const int32_t* mir, int64_t num_unique, const float* w1,
                         const float* w2, const float* w3, const float* w4, const float* gout, float* dw1, float* dw2, float* dw3,
                         float* dw4, int64_t num_edges, int32_t S, int32_t Sout, void* ws, size_t ws_bytes, gml_stream_t stream);
size_t gml_edge_mlp_bwd_workspace_bytes(int64_t num_edges, int32_t S, int32_t Sout);
/* gout: dL/dout [E, Sout].  Writes dw1..dw4 (same shapes as the weights) and, if gin != NULL,
 * dL/dea [E, S].  Intermediates are recomputed from ea. */
int gml_edge_mlp_bwd(const float* ea, const void* ea_split, const float* w1, const float* w2, const float* w3,
                     const float* w4, const float* gout, float* gin, float* dw1, float* dw2, float* dw3, float* dw4,
                     int64_t num_edges, int32_t S, int32_t Sout,
                     void* ws, size_t ws_bytes, gml_stream_t stream);
/* the edge branch on the exact-arithmetic family whatever the shape -- one edge per lane, fp32 FMAs, f32-input MFMA for the weight
 * gradients, the library's tanh: what GML_F32_MFMA is to gml_spectconv_fwd / _bwd (the default entries above take the bf16-split
 * matrix-core chains for 2 <= S <= 16).  Same arguments minus the pre-split image; the backward's dw1 .. dw4 are required. */
int gml_edge_mlp_fwd_exact(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4, float* out,
                           const int32_t* tpos, float* out_t, int64_t num_edges, int32_t S, int32_t Sout, gml_stream_t stream);
int gml_edge_mlp_bwd_exact(const float* ea, const float* w1, const float* w2, const float* w3, const float* w4,
                           const float* gout, float* gin, float* dw1, float* dw2, float* dw3, float* dw4,
                           int64_t num_edges, int32_t S, int32_t Sout, void* ws, size_t ws_bytes, gml_stream_t stream);

/* ---------------------------------------------------------------- ML3Layer forward without the edge branch
 * (libs/spect_conv.py:204-212): out[:, :nout1] = act(SpectConv(x)), out[:, nout1:nout1+F2] = tanh(fc11 x) * tanh(fc12 x).
 * Same arguments as gml_spectconv_fwd + the Hadamard weights; one launch on the 8-wave kernel when GML_GROUPS128 applies
 * and F2 <= 8, else gml_spectconv_fwd followed by gml_node_mix_fwd.  F2 = 0: conv only.
 * epos != NULL: the value row of CSR position k is val[epos[k]] -- with epos = the target-to-source position map the
 * layer consumes the edge branch's output in SOURCE order (the order the backward walks), so the edge branch writes its
 * output once instead of once per order. */
int gml_ml3_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* ginfo, const int32_t* epos, const float* val,
                const float* x, int64_t ldx, const float* w, int64_t w_ss, int64_t w_si, int64_t w_so,
                const float* bias, const float* w11, const float* b11, const float* w12, const float* b12,
                float* out, int64_t ldo, int64_t num_rows, int32_t S, int32_t Fin, int32_t nout1, int32_t F2,
                uint32_t flags, gml_stream_t stream);

/* ---------------------------------------------------------------- ML3Layer Hadamard branch
 *   out[r, 0:F2] = tanh(x[r] . w11^T + b11) * tanh(x[r] . w12^T + b12)      w1x: [F2, Fin] */
int gml_node_mix_fwd(const float* x, int64_t ldx, const float* w11, const float* b11,
                     const float* w12, const float* b12, float* out, int64_t ldo,
                     int64_t num_rows, int32_t Fin, int32_t F2, gml_stream_t stream);
/* Backward, one launch + fold: with z11 = x w11^T + b11, z12 likewise (recomputed from x),
 *   dx[r, :] += dL/dz11[r] . w11 + dL/dz12[r] . w12        (accumulates; dx may be NULL)
 *   dw11 = dL/dz11^T x, db11 = colsum(dL/dz11), dw12, db12 likewise.
 * gout: dL/dout with leading dimension ldg.  workspace_bytes == 0: shape not supported (caller uses GEMMs). */
size_t gml_node_mix_bwd_workspace_bytes(int64_t num_rows, int32_t Fin, int32_t F2);
int gml_node_mix_bwd(const float* x, int64_t ldx, const float* w11, const float* b11,
                     const float* w12, const float* b12, const float* gout, int64_t ldg,
                     float* dx, int64_t lddx, float* dw11, float* db11, float* dw12, float* db12,
                     int64_t num_rows, int32_t Fin, int32_t F2, void* ws, size_t ws_bytes, gml_stream_t stream);

/* ---------------------------------------------------------------- ML3Layer output stage, backward in one pass
 * (autograd of libs/spect_conv.py:209-212: the relu on the conv part, the concat, the Hadamard branch, conv1.bias)
 *   G[:, :nout1] = gy[:, :nout1] * (y[:, :nout1] > 0), G[:, nout1:ldg] = 0      gradient at the conv output
 *   dcb          = column sums of G                                             (NULL: not wanted)
 *   dx           = dz11 w11 + dz12 w12  -- WRITTEN (follow with gml_spectconv_bwd + GML_ACCUM); NULL: not wanted
 *   dw11, db11, dw12, db12 as gml_node_mix_bwd, from gy[:, nout1:nout1+F2].
 * F2 == 0: no Hadamard branch (x, w*, dw*, dx ignored) -- relu mask + bias sums of a plain SpectConv.
 * workspace_bytes == 0: shape not supported (caller uses gml_relu_bwd + gml_node_mix_bwd). */
size_t gml_ml3_split_bwd_workspace_bytes(int64_t num_rows, int32_t Fin, int32_t nout1, int32_t F2);
int gml_ml3_split_bwd(const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                      const float* w11, const float* b11, const float* w12, const float* b12,
                      float* G, int64_t ldg, float* dx, int64_t lddx, float* dcb,
                      float* dw11, float* db11, float* dw12, float* db12,
                      int64_t num_rows, int32_t Fin, int32_t nout1, int32_t F2,
                      void* ws, size_t ws_bytes, gml_stream_t stream);

/* general form.  dz != NULL (then dx == NULL; 2 F2 <= 4): dz [num_rows, 4] = (dz11 | dz12) is written instead of dx (columns
 * beyond 2 F2 zero): the operand of gml_spectconv_bwd_mix.  gy_seg != NULL: gy holds one row per SEGMENT and row r reads
 * gy[gy_seg[r]] -- the gradient of a global add pool that directly follows the layer (Zinc12k.py:343), never expanded.
 * y == NULL and G == NULL (pre-masked form; gy_seg NULL): gy[:, :nout1] already carries the relu mask
 * (gml_spectconv_bwd_mix_relu wrote it): the saved output is not read and no G is written -- gy is G. */
int gml_ml3_split_bwd_ex(const float* gy, int64_t ldgy, const int32_t* gy_seg, const float* y, int64_t ldy,
                         const float* x, int64_t ldx, const float* w11, const float* b11, const float* w12, const float* b12,
                         float* G, int64_t ldg, float* dx, int64_t lddx, float* dz, float* dcb,
                         float* dw11, float* db11, float* dw12, float* db12,
                         int64_t num_rows, int32_t Fin, int32_t nout1, int32_t F2,
                         void* ws, size_t ws_bytes, gml_stream_t stream);

/* ---------------------------------------------------------------- glue
 * g[r, c] = (y[r, c] > 0) ? gy[r, c] : 0   for c < F, and 0 for F <= c < ldg (relu backward on a strided
 * slice; the zero padding lets consumers read aligned float4 groups) */
int gml_relu_bwd(const float* gy, int64_t ldgy, const float* y, int64_t ldy, float* g, int64_t ldg,
                 int64_t num_rows, int32_t F, gml_stream_t stream);
/* out[g, :] = sum_{r in [ptr[g], ptr[g+1])} x[r, :]  (global_add_pool over a sorted batch vector;
 * mean bit 0 divides by the segment length: global_mean_pool; bit 1 = GML_POOL_SKIP_LAST: the last segment is the padding graph of a
 * static-shape batch (gml_batch_assemble) -- its output row is written as zeros without reading its rows) */
#define GML_POOL_SKIP_LAST 2
int gml_segment_sum(const float* x, int64_t ldx, const int32_t* ptr, float* out, int64_t ldo,
                    int64_t num_segments, int32_t F, int32_t mean, gml_stream_t stream);
/* its gradient: out[r, :] = g[seg(r), :] (/ segment length when mean != 0) for r in [ptr[seg], ptr[seg+1]) */
int gml_segment_bcast(const float* g, int64_t ldg, const int32_t* ptr, float* out, int64_t ldo,
                      int64_t num_segments, int32_t F, int32_t mean, gml_stream_t stream);

/* gml_segment_sum for 32-column rows that also leaves the relu pattern of every row -- bit c of mask[row] = (x[row][c] > 0) -- and
 * the pool gradient broadcast that applies it: out[row][c] = g[seg[row]][c] * (c >= nrelu || bit c of mask[row]).  For the ML3Layer
 * directly in front of global_add_pool / global_mean_pool (Zinc12k.py:338-343, out = [relu(conv) | Hadamard columns]): its backward
 * reads 4 bytes per row instead of the saved output, and the pre-masked [num_rows, 32] gradient is what gml_spectconv_bwd_had takes.
 * F must be 32 (GML_E_UNSUPPORTED otherwise); mean: the flag word of gml_segment_sum; for a mean pool the caller divides g by the
 * segment sizes first.  Same sums, in the same order, as gml_segment_sum. */
int gml_segment_sum_mask(const float* x, int64_t ldx, const int32_t* ptr, float* out, int64_t ldo, uint32_t* mask,
                         int64_t num_segments, int32_t F, int32_t mean, gml_stream_t stream);
int gml_segment_bcast_mask(const float* g, int64_t ldg, const int32_t* seg, const uint32_t* mask, float* out, int64_t ldo,
                           int64_t num_rows, int32_t F, int32_t nrelu, gml_stream_t stream);
/* out[g, c] = max_{r in [ptr[g], ptr[g+1])} x[r, c]  (global_max_pool, /root/reference/enzymes.py:384); argmax[g, c]
 * (int32 [num_segments, F], may be NULL) = the first row that attains it; an empty segment gives 0 / -1 */
int gml_segment_max(const float* x, int64_t ldx, const int32_t* ptr, float* out, int64_t ldo, int32_t* argmax,
                    int64_t num_segments, int32_t F, gml_stream_t stream);
/* its gradient: out[r, c] = (r == argmax[seg(r), c]) ? g[seg(r), c] : 0 -- every row of every segment is written */
int gml_segment_max_bwd(const float* g, int64_t ldg, const int32_t* ptr, const int32_t* argmax, float* out, int64_t ldo,
                        int64_t num_segments, int32_t F, gml_stream_t stream);

/* ---------------------------------------------------------------- dense-block SpectConv (equal-size graphs, near-dense masks)
 * The TF formulation of the layer, /root/reference/libs/layers_tf.py:231-236 (s0 = matmul(support[:, i], x); out += s0 . W_i),
 * for batches of B graphs of exactly n <= 96 nodes (MNIST-75: n = 75, S = 6).
 *   gml_dense_pack       : fp32 blocks [nblocks][n][n] (row-major; transpose != 0 takes block^T) -> bf16 (hi, lo) images
 *                          img [nblocks][2][n][KP] (uint16; block = hi + lo to 2^-17 relative; KP = 32 ceil(n / 32), columns
 *                          >= n zero).  Done once per data set: the supports are constants.
 *   gml_dense_support_mm : out[(b n + r) ldo + s so + f]  =  sum_k D[b][s][r][k] . act[(b n + k) lda + s sa + f]   (f < F <= 128),
 *                          summed over s into out[(b n + r) ldo + f] when sum_s != 0.  Forward: D = the packed supports,
 *                          act = X (sa = 0), out = Hcat [B n, S Fin] (so = Fin); backward: D = the packed transposes,
 *                          act = d Hcat (sa = Fin), sum_s = 1 -> d X.  bf16x3 products, fp32 accumulate. */
int gml_dense_pack(const float* blocks, uint16_t* img, int64_t nblocks, int32_t n, int32_t KP, int32_t transpose,
                   gml_stream_t stream);
int gml_dense_support_mm(const uint16_t* dimg, const float* act, int64_t lda, int32_t sa, float* out, int64_t ldo,
                         int32_t so, int32_t sum_s, int32_t B, int32_t S, int32_t n, int32_t KP, int32_t F,
                         gml_stream_t stream);
/* The layer in ONE launch, as libs/layers_tf.py:231-236 forms it (matmul(support, x) then the projection by W_i): the support
 * product's accumulators are the projection's operand, Hcat never goes through a library GEMM.
 *   gml_dense_pack_w   : weight [S][Fin][Fout] fp32 -> bf16 (hi, lo) projection fragments (gml_dense_wimg_elems int16 elements)
 *   gml_dense_conv_fwd : out2[(b n + r) ldo2 + o] = act?( sum_s sum_f (sum_k D[b][s][r][k] x[(b n + k) ldx + f]) W[s][f][o] + bias[o] );
 *                        hcat (optional, [B n, S Fin] contiguous) receives D_s X for the caller's weight gradient. Fin, Fout <= 128. */
size_t gml_dense_wimg_elems(int32_t S, int32_t Fin, int32_t Fout);
int gml_dense_pack_w(const float* w, uint16_t* wimg, int32_t S, int32_t Fin, int32_t Fout, void* stream);
int gml_dense_conv_fwd(const uint16_t* dimg, const float* x, int64_t ldx, const uint16_t* wimg, const float* bias,
                       float* out2, int64_t ldo2, float* hcat, int32_t B, int32_t S, int32_t n, int32_t KP, int32_t Fin,
                       int32_t Fout, int32_t relu, void* stream);
/* and its gradient w.r.t. x in one launch: dx[(b n + i) lddx + f] = sum_s sum_j D[b][s][j][i] (sum_o g[(b n + j) ldg + o] W[s][f][o])
 * -- the projection g W_s^T per row tile, then the support product on the TRANSPOSED packed blocks (gml_dense_pack(transpose = 1));
 * d Hcat is never materialised.  wimgT: gml_dense_pack_wt (gml_dense_wimgt_elems int16 elements). */
size_t gml_dense_wimgt_elems(int32_t S, int32_t Fin, int32_t Fout);
int gml_dense_pack_wt(const float* w, uint16_t* wimgT, int32_t S, int32_t Fin, int32_t Fout, void* stream);
int gml_dense_conv_bwd_x(const uint16_t* dimgT, const float* g, int64_t ldg, const uint16_t* wimgT, float* dx, int64_t lddx,
                         int32_t B, int32_t S, int32_t n, int32_t KP, int32_t Fin, int32_t Fout, void* stream);
/* the layer's weight gradient WITHOUT Hcat and without a library GEMM: dW[s][f][o] = sum_b sum_j (D[b][s] X[b])[j][f] g[b n + j][o] -- the
 * support product recomputed per graph on the matrix cores and contracted with g over the graph's rows (K = 16 MFMAs on the
 * untransposed product), per-slice partial sums in ws ([gml_dense_dw_slices(B)][S][Fin][Fout]) folded in slice order into dw
 * (dw == NULL: the partials stay for gml_fold_many).  dimg: the forward images (gml_dense_pack(transpose = 0)); Fin <= 128. */
int32_t gml_dense_dw_slices(int32_t B);
size_t gml_dense_dw_workspace_bytes(int32_t B, int32_t S, int32_t Fin, int32_t Fout);
int gml_dense_conv_bwd_w(const uint16_t* dimg, const float* x, int64_t ldx, const float* g, int64_t ldg, float* dw,
                         int32_t B, int32_t S, int32_t n, int32_t KP, int32_t Fin, int32_t Fout, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- support precompute on the device (adjacent step, P1)
 * SpectralDesign.__call__ (libs/utils.py:546-610) for a batch of graphs with at most 80 nodes each (larger: host
 * implementation).  node_ptr [B+1], edge_ptr [B+1]: graph b owns nodes [node_ptr[b], node_ptr[b+1]) and the edges
 * [edge_ptr[b], edge_ptr[b+1]) of edge_index [2, e_total] (int64, global node ids).
 *   gml_spectral_count : nnz[b] = number of entries of the mask M_b (A or (A+I) squared recfield-1 times, > 0);
 *   gml_spectral_design: with out_ptr = exclusive scan of nnz ([B+1], int64) writes, at out_ptr[b], the row-major COO of
 *     M_b (edge_index2 [2, m_total] int64, global ids) and per entry the S = nfreq + 1 (+1) support values
 *     (edge_attr2 [m_total, S]: nfreq Gaussian band-pass filters of the normalised-Laplacian spectrum -- or of A's when
 *     laplacien == 0 --, the identity, and A when addadj), plus lmax[b] = largest Laplacian eigenvalue.
 * Eigenproblems in float64 (cyclic Jacobi in LDS); has_vmax/vmax = fixed upper end of the frequency grid. */
int gml_spectral_count(const int32_t* node_ptr, const int32_t* edge_ptr, const int64_t* edge_index, int64_t e_total,
                       int64_t num_graphs, int32_t max_nodes, int32_t recfield, int32_t* nnz, gml_stream_t stream);
int gml_spectral_design(const int32_t* node_ptr, const int32_t* edge_ptr, const int64_t* edge_index, int64_t e_total,
                        int64_t num_graphs, int32_t max_nodes, int32_t recfield, int32_t nfreq, double dv,
                        int32_t has_vmax, double vmax, int32_t laplacien, int32_t addadj, const int64_t* out_ptr,
                        int64_t m_total, int64_t* edge_index2, float* edge_attr2, float* lmax, gml_stream_t stream);

/* torch.nn.BatchNorm1d in training mode over the rows of x [num_rows, C] (mutag.py:272-288: one between every two layers): batch
 * statistics (mean, BIASED variance, rstd = 1 / sqrt(var + eps)), y = (x - mean) rstd weight + bias, and the backward: sum_dy = d bias,
 * sum_dyxhat = d weight, dx = (dy - sum_dy / n - xhat sum_dyxhat / n) rstd weight.  C <= 64, C % 4 == 0, float4-addressable rows; other
 * shapes: GML_E_UNSUPPORTED.  weight / bias may be NULL (affine=False).  ws: gml_bn_workspace_bytes(num_rows). */
size_t gml_bn_workspace_bytes(int64_t num_rows);
int gml_bn_stats(const float* x, int64_t ldx, int64_t num_rows, int32_t C, float eps, float* mean, float* var, float* rstd,
                 void* ws, size_t ws_bytes, gml_stream_t stream);
int gml_bn_apply(const float* x, int64_t ldx, int64_t num_rows, int32_t C, const float* mean, const float* rstd, const float* weight,
                 const float* bias, float* y, int64_t ldy, gml_stream_t stream);
int gml_bn_bwd_sums(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t num_rows, int32_t C, const float* mean,
                    const float* rstd, float* sum_dy, float* sum_dyxhat, void* ws, size_t ws_bytes, gml_stream_t stream);
int gml_bn_bwd_apply(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t num_rows, int32_t C, const float* mean,
                     const float* rstd, const float* weight, const float* sum_dy, const float* sum_dyxhat, float* dx, int64_t lddx,
                     gml_stream_t stream);

/* out[i, j] = sum_r A[r, i] * B[r, j]  (a, b <= 64): weight gradient g^T x of a small dense layer over n rows
 * (readout head fc1 / fc2, Zinc12k.py:343-345), rows split over the chip, fixed summation order */
size_t gml_xty_workspace_bytes(int64_t n, int32_t a, int32_t b);
int gml_xty(const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t n, int32_t a, int32_t b,
            void* ws, size_t ws_bytes, gml_stream_t stream);

/* out[i, j] = sum_r A[r, i] * B[r, j] for a wide A (a <= 4096 columns) and b <= 128: the dense-block layer's weight gradient
 * dW = Hcat^T g (libs/layers_tf.py:231-236, its autograd: a = S Fin, b = Fout) on the bf16 matrix cores (bf16x3 split products, fp32
 * accumulate), rows along K through transposing LDS reads; per-row-range partials folded in order.  ws: gml_xty_wide_workspace_bytes. */
int gml_xty_wide_supported(int64_t n, int32_t a, int32_t b);
size_t gml_xty_wide_workspace_bytes(int64_t n, int32_t a, int32_t b);
int gml_xty_wide(const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t n, int32_t a, int32_t b,
                 void* ws, size_t ws_bytes, gml_stream_t stream);

/* Readout head + L1-sum loss of the ZINC GNNML3 in one launch each way (Zinc12k.py:343-345, :365) for the reference's regime, a
 * batch of 64 graphs, where head, loss and their backward were ~20 of the step's 70 launches (csrc/gml_head.hip):
 *   loss[0] = sum_{r < rows_loss} valid[r] |w2 . relu(W1 p[r] + b1) + b2 - y[r]|     p [rows, nin] pooled features, W1 [nh, nin]
 * rows >= rows_loss (the padding graph of a static batch) enter neither the loss nor any gradient.  pre (optional): logits of all
 * rows.  The backward recomputes from p and scales by gscale[0] (device scalar, NULL = 1): gp [rows, nin], dw1 [nh, nin], db1 [nh]
 * (NULL allowed), dw2 [nh], db2 [1] (NULL allowed).  One workgroup: rows <= 256, nin, nh <= 64 and everything in LDS, else
 * GML_E_UNSUPPORTED (large batches run the general path: library GEMMs + gml_xty). */
int gml_head_l1_fwd(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                    const float* w2, const float* b2, int32_t rows, int32_t rows_loss, int32_t nin, int32_t nh,
                    float* loss, float* pre, gml_stream_t stream);
/* gml_head_l1_fwd + loss_sum[0] += loss (NULL: none): the epoch's running loss (Zinc12k.py:366) without a launch of its own */
int gml_head_l1_fwd_acc(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                        const float* w2, const float* b2, int32_t rows, int32_t rows_loss, int32_t nin, int32_t nh,
                        float* loss, float* loss_sum, float* pre, gml_stream_t stream);
int gml_head_l1_bwd(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                    const float* w2, const float* b2, int32_t rows, int32_t rows_loss, int32_t nin, int32_t nh,
                    const float* gscale, float* gp, int64_t ldgp, float* dw1, float* db1, float* dw2, float* db2,
                    gml_stream_t stream);

/* The same head + loss for ANY number of rows (nin = nh = 32: GNNML3's 30 + 2 features, fc1 32 -> 32; GML_E_UNSUPPORTED otherwise --
 * the caller keeps its general road): one pass forward + the fold of the per-workgroup loss partials, one pass backward (forward
 * recomputed; gp written; dw1 / db1 / dw2 / db2 contracted over the rows with exact f32 matrix-core products, one partial per
 * workgroup) + the fixed-order fold.  ws: gml_head_l1_big_workspace_floats(rows, nin, nh) floats (0: shape not served).  _bwd with
 * dw1 = dw2 = NULL: the partials [parts][1089] (dw1 | db1 | dw2 | db2) stay in ws for gml_fold_many, parts = workspace_floats / 1089.
 * Replaces ~16 launches of the general road (Zinc12k.py:343-345, :365 at the bench's batch size). */
size_t gml_head_l1_big_workspace_floats(int64_t rows, int32_t nin, int32_t nh);
int gml_head_l1_big_fwd(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                        const float* w2, const float* b2, int64_t rows, int64_t rows_loss, int32_t nin, int32_t nh,
                        float* loss, float* loss_sum, void* ws, size_t ws_floats, gml_stream_t stream);
int gml_head_l1_big_bwd(const float* p, int64_t ldp, const float* y, const float* valid, const float* w1, const float* b1,
                        const float* w2, const float* b2, int64_t rows, int64_t rows_loss, int32_t nin, int32_t nh,
                        const float* gscale, float* gp, int64_t ldgp, float* dw1, float* db1, float* dw2, float* db2,
                        void* ws, size_t ws_floats, gml_stream_t stream);

/* GNNML1 block in one launch each way (csrc/gml_gnnml1.hip) -- /root/reference/sr25.py:231-240 (graph8c.py: the same class),
 * mnist75.py:296-318, mutag.py:253-262.  a = fc_i1(x), c = conv_i1(x) = (A^T x) Wc + bc (SpectConv, K = 1, selfconn = False,
 * libs/spect_conv.py:64-96 with one support), f2 = fc_i2(x), f3 = fc_i3(x):
 *   mode 0: out [N, n] = act(a + c + f2 * f3)   (n1 = n2 = n3 = n)        mode 1: out [N, n1 + n2 + n3] = [act(a) | act(c) | act(f2 * f3)]
 *   mode 2: out = [act(a) | act(c) | act(f2) * act(f3)]  (mutag.py)       act: 0 = tanh, 1 = relu
 * w1, w2, w3: Linear weights [n, Fin] row-major, b*: [n] or NULL; wc: the SpectConv weight [Fin, n2]; val: one value per edge in the
 * order of `col` (NULL: ones, the scripts' torch.ones edge_attr).  Exact fp32 products.  Fin, n1, n2, n3 <= 64 (gml_gnnml1_supported),
 * else GML_E_UNSUPPORTED and the caller composes the block from gml_spectconv_fwd and library Linears.
 * Backward (source-keyed view rowptr_t / col_t / val_t; `out` = the saved forward output, gout = dL/dout):
 *   dx (optional) = da W1 + df2 W2 + df3 W3 + (A dc) Wc^T                    da, dc, df2, df3 = gradients at a, c, f2, f3
 *   g4 [N, ldg4 >= gml_gnnml1_g4_cols()] = [da | dc (modes 1, 2 only: dc = da in mode 0) | df2 | df3], each block 16 ceil(n / 16) columns wide
 *   q  [N, ldq >= 16 ceil(n2 / 16)]      = A dc
 * from which the caller forms dW1 = da^T x, dW2 = df2^T x, dW3 = df3^T x, dWc = x^T q (gml_xty) and the bias gradients (column sums). */
int gml_gnnml1_supported(int32_t Fin, int32_t n1, int32_t n2, int32_t n3, int32_t mode);
int gml_gnnml1_g4_cols(int32_t n1, int32_t n2, int32_t n3, int32_t mode);
int gml_gnnml1_fwd(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t ldx, int64_t num_rows,
                   int32_t Fin, const float* w1, const float* b1, int32_t n1, const float* wc, const float* bc, int32_t n2,
                   const float* w2, const float* b2, const float* w3, const float* b3, int32_t n3, int32_t mode, int32_t act,
                   float* out, int64_t ldo, gml_stream_t stream);
/* the block's weight and bias gradients from g4 / q in one pass over the rows + one fold:
 *   out_flat = [dW1 (n1 x Fin) | dW2 (n3 x Fin) | dW3 (n3 x Fin) | dWc (Fin x n2) | column sums of g4 (gml_gnnml1_g4_cols floats)]
 * (db1 = sums of the da block, dbc = sums of the dc block -- the da block in mode 0 --, db2 / db3 = sums of the df2 / df3 blocks) */
int64_t gml_gnnml1_dw_floats(int32_t Fin, int32_t n1, int32_t n2, int32_t n3, int32_t mode);
size_t gml_gnnml1_dw_workspace_bytes(int64_t num_rows, int32_t Fin, int32_t n1, int32_t n2, int32_t n3, int32_t mode);
int gml_gnnml1_dw(const float* x, int64_t ldx, const float* g4, int64_t ldg4, const float* q, int64_t ldq, int64_t num_rows,
                  int32_t Fin, int32_t n1, int32_t n2, int32_t n3, int32_t mode, float* out_flat, void* ws, size_t ws_bytes,
                  gml_stream_t stream);
int gml_gnnml1_bwd(const int32_t* rowptr_t, const int32_t* col_t, const float* val_t, const float* x, int64_t ldx,
                   const float* out, int64_t ldo, const float* gout, int64_t ldgo, int64_t num_rows, int32_t Fin,
                   const float* w1, int32_t n1, const float* wc, int32_t n2, const float* w2, const float* b2, const float* w3,
                   const float* b3, int32_t n3, int32_t mode, int32_t act, float* dx, int64_t lddx, float* g4, int64_t ldg4,
                   float* q, int64_t ldq, gml_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GML_H_ */
