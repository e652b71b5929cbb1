// COO -> CSR on the device, integer-exact and deterministic.
//
// The reference never builds a CSR: edge_index [2,E] goes to PyG propagate whose CPU scatter-add
// sums the messages of one target in ascending edge order (/root/reference/libs/spect_conv.py:77).
// To walk rows instead of scattering, edges are bucketed by key (histogram + scan + atomic slots)
// and every bucket is then sorted by input edge id, which makes the result equal to a STABLE sort
// by key whatever order the atomics retired in (oracle: oracle/csr_oracle.py).
#include "gml_common.h"

#define SCAN_BLOCK 256
#define SCAN_ITEMS 4
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

// A key outside [0, N) is clamped (so nothing is written outside the arrays) and reported through *bad: the caller
// raises after its next host read -- the reference's scatter raises an index error for the same input.
__device__ __forceinline__ int64_t gml_checked_key(int64_t k, int64_t N, int32_t* __restrict__ bad) {
    if ((uint64_t)k >= (uint64_t)N) {
        atomicOr(bad, 1);
        return k < 0 ? 0 : N - 1;
    }
    return k;
}

__global__ void gml_k_hist(const int64_t* __restrict__ key, int64_t E, int64_t N, int32_t* __restrict__ counts1,
                           int32_t* __restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) atomicAdd(&counts1[gml_checked_key(key[e], N, bad) + 1], 1);
}

// in-place inclusive scan of tiles; tile totals -> sums[b]
__global__ __launch_bounds__(SCAN_BLOCK) void gml_k_scan_tiles(int32_t* __restrict__ a, int64_t n,
                                                              int32_t* __restrict__ sums) {
    __shared__ int32_t wsum[SCAN_BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int32_t v[SCAN_ITEMS];
    int32_t run = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? a[base + i] : 0;
        run += v[i];
        v[i] = run;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t incl = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int32_t excl = woff + incl - run;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) a[base + i] = v[i] + excl;
    if (threadIdx.x == SCAN_BLOCK - 1) sums[blockIdx.x] = woff + incl;
}

// exclusive scan of the tile totals by one workgroup (serial over chunks, carry kept in a register)
__global__ __launch_bounds__(SCAN_BLOCK) void gml_k_scan_sums(int32_t* __restrict__ sums, int64_t nb) {
    __shared__ int32_t wsum[SCAN_BLOCK / 64];
    __shared__ int32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t c0 = 0; c0 < nb; c0 += SCAN_BLOCK) {
        const int64_t i = c0 + threadIdx.x;
        const int32_t x = (i < nb) ? sums[i] : 0;
        int32_t incl = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int32_t woff = carry_s;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        if (i < nb) sums[i] = woff + incl - x;
        __syncthreads();
        if (threadIdx.x == SCAN_BLOCK - 1) carry_s = woff + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void gml_k_scan_add(int32_t* __restrict__ a, int64_t n,
                                                            const int32_t* __restrict__ sums) {
    const int32_t off = sums[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) a[base + i] += off;
}

__global__ void gml_k_slot(const int64_t* __restrict__ key, int64_t E, int64_t N, int32_t* __restrict__ cursor,
                           int32_t* __restrict__ perm, int32_t* __restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) perm[atomicAdd(&cursor[gml_checked_key(key[e], N, bad)], 1)] = (int32_t)e;
}

// one lane per row: insertion sort of the row's edge ids (rows are short and nearly sorted)
__global__ void gml_k_sort_rows(const int32_t* __restrict__ rowptr, int64_t N, int32_t* __restrict__ perm) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const int b = rowptr[r], e = rowptr[r + 1];
    for (int i = b + 1; i < e; ++i) {
        const int32_t v = perm[i];
        int j = i - 1;
        while (j >= b && perm[j] > v) { perm[j + 1] = perm[j]; --j; }
        perm[j + 1] = v;
    }
}

__global__ void gml_k_take_i64(const int64_t* __restrict__ in, const int32_t* __restrict__ perm, int64_t E, int64_t N,
                               int32_t* __restrict__ out, int32_t* __restrict__ bad) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < E) out[k] = (int32_t)gml_checked_key(in[perm[k]], N, bad);
}

extern "C" size_t gml_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges) {
    (void)num_edges;
    if (num_nodes < 0) return 0;
    const int64_t nb = gml_cdiv(num_nodes + 1, SCAN_TILE);
    return (size_t)(nb + 1 + num_nodes + 1 + 1) * sizeof(int32_t);     /* scan sums, cursors, the bad-id flag (last int32) */
}

extern "C" int gml_csr_from_coo(const int64_t* key, const int64_t* other_in, int64_t num_nodes, int64_t num_edges,
                                int32_t* rowptr, int32_t* other, int32_t* perm, void* ws, size_t ws_bytes,
                                gml_stream_t stream) {
    if (num_nodes < 0 || num_edges < 0) return GML_E_BADARG;
    if (num_nodes >= INT32_MAX || num_edges >= INT32_MAX) return GML_E_UNSUPPORTED;
    if (!rowptr) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n1 = num_nodes + 1;
    gml_zero_async(rowptr, sizeof(int32_t) * n1, st);
    if (num_edges == 0 || num_nodes == 0) return GML_OK;
    if (!key || !other_in || !other || !perm || !ws) return GML_E_BADARG;
    if (ws_bytes < gml_csr_workspace_bytes(num_nodes, num_edges)) return GML_E_WORKSPACE;
    const int64_t nb = gml_cdiv(n1, SCAN_TILE);
    int32_t* sums = (int32_t*)ws;
    int32_t* cursor = sums + nb + 1;
    int32_t* bad = cursor + n1;                                /* OR-ed, never cleared here: the caller zeroes it */
    const unsigned eg = (unsigned)gml_cdiv(num_edges, 256);

    hipLaunchKernelGGL(gml_k_hist, dim3(eg), dim3(256), 0, st, key, num_edges, num_nodes, rowptr, bad);
    hipLaunchKernelGGL(gml_k_scan_tiles, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, rowptr, n1, sums);
    hipLaunchKernelGGL(gml_k_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, st, sums, nb);
    hipLaunchKernelGGL(gml_k_scan_add, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, rowptr, n1, sums);
    gml_copy_async(cursor, rowptr, sizeof(int32_t) * n1, st);
    hipLaunchKernelGGL(gml_k_slot, dim3(eg), dim3(256), 0, st, key, num_edges, num_nodes, cursor, perm, bad);
    hipLaunchKernelGGL(gml_k_sort_rows, dim3((unsigned)gml_cdiv(num_nodes, 256)), dim3(256), 0, st, rowptr,
                       num_nodes, perm);
    hipLaunchKernelGGL(gml_k_take_i64, dim3(eg), dim3(256), 0, st, other_in, perm, num_edges, num_nodes, other, bad);
    return gml_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Keys that are ALREADY non-decreasing (the reference's transform emits edge_index2 in row-major np.where order,
// libs/utils.py:608-609: sorted by source): the view keyed by them needs no sort at all -- rowptr from the run
// boundaries, perm = identity.  A decreasing pair sets bit 1 of *bad (the caller then builds this view the general way).
__global__ void gml_k_sorted_view(const int64_t* __restrict__ key, const int64_t* __restrict__ other_in, int64_t E, int64_t N,
                                  int32_t* __restrict__ rowptr, int32_t* __restrict__ other, int32_t* __restrict__ perm,
                                  int32_t* __restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > E) return;
    if (e < E) {
        other[e] = (int32_t)gml_checked_key(other_in[e], N, bad);
        perm[e] = (int32_t)e;
    }
    const int64_t kprev = e == 0 ? -1 : gml_checked_key(key[e - 1], N, bad);
    const int64_t kcur = e == E ? N : gml_checked_key(key[e], N, bad);
    // Not sorted: flagged.  rowptr was zero-filled and only ever receives values <= E, perm is the identity and other is
    // range-checked, so whatever the caller launches on these arrays before it reads the flag stays inside them.
    if (kcur < kprev) { atomicOr(bad, 2); return; }
    for (int64_t r = kprev + 1; r <= kcur; ++r) rowptr[r] = (int32_t)e;      // rows (kprev, kcur] start at e (empty rows included)
}

extern "C" int gml_csr_from_sorted_coo(const int64_t* key, const int64_t* other_in, int64_t num_nodes, int64_t num_edges,
                                       int32_t* rowptr, int32_t* other, int32_t* perm, void* ws, size_t ws_bytes,
                                       gml_stream_t stream) {
    if (num_nodes < 0 || num_edges < 0) return GML_E_BADARG;
    if (num_nodes >= INT32_MAX || num_edges >= INT32_MAX) return GML_E_UNSUPPORTED;
    if (!rowptr) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    gml_zero_async(rowptr, sizeof(int32_t) * (num_nodes + 1), st);
    if (num_edges == 0 || num_nodes == 0) return GML_OK;
    if (!key || !other_in || !other || !perm || !ws) return GML_E_BADARG;
    if (ws_bytes < gml_csr_workspace_bytes(num_nodes, num_edges)) return GML_E_WORKSPACE;
    int32_t* bad = (int32_t*)((char*)ws + gml_csr_workspace_bytes(num_nodes, num_edges) - sizeof(int32_t));
    hipLaunchKernelGGL(gml_k_sorted_view, dim3((unsigned)gml_cdiv(num_edges + 1, 256)), dim3(256), 0, st, key, other_in,
                       num_edges, num_nodes, rowptr, other, perm, bad);
    return gml_launch_status();
}

__global__ void gml_k_invert(const int32_t* __restrict__ perm, int64_t E, int32_t* __restrict__ inv) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < E) inv[perm[k]] = (int32_t)k;
}

__global__ void gml_k_take_i32(const int32_t* __restrict__ in, const int32_t* __restrict__ perm, int64_t E,
                               int32_t* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < E) out[k] = in[perm[k]];
}

extern "C" int gml_csr_link_transpose(const int32_t* perm_fwd, const int32_t* perm_t, int64_t num_edges,
                                      int32_t* inv_scratch, int32_t* pos_t, gml_stream_t stream) {
    if (num_edges < 0) return GML_E_BADARG;
    if (num_edges == 0) return GML_OK;
    if (!perm_fwd || !perm_t || !inv_scratch || !pos_t) return GML_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned eg = (unsigned)gml_cdiv(num_edges, 256);
    hipLaunchKernelGGL(gml_k_invert, dim3(eg), dim3(256), 0, st, perm_fwd, num_edges, inv_scratch);
    hipLaunchKernelGGL(gml_k_take_i32, dim3(eg), dim3(256), 0, st, inv_scratch, perm_t, num_edges, pos_t);
    return gml_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Per group of 64 (or 128) rows one record of GML_GREC_INTS(group_rows) ints:
//   {first edge, #edges, smallest column id, width of the column window}  -- the fused layer kernels read these 16
//   bytes and know at once which CSR slice / value rows / X rows to prefetch (no dependent rowptr -> col -> min/max
//   chain inside the hot kernel);
//   128-row groups only: order[128] bytes -- the row (local index) each lane position of the consuming kernel works on.  Rows are
//   ranked by degree (descending, ties by index) so the 16 rows of a tile run near-equal edge loops (lane utilisation
//   of the edge phase 0.59 -> 0.90 on ZINC-like supports); rank blocks are dealt to the waves so that the SIMDs
//   stay balanced: 128-row groups (backward, 8 waves, waves w and w+4 share a SIMD) pair block a with block 7-a
//   and rotate a with the group index.  64-row groups (forward: 4 waves, one tile per SIMD, a barrier per group)
//   carry no order: concentrating the long rows in one tile lengthens the group's critical path by more than the
//   shorter tiles save (measured +2.6 %).
//   Which row a lane serves does not change any row's result (each row keeps its own edge order).
__global__ __launch_bounds__(128) void gml_k_group_info(const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, int64_t nrows,
                                                       int32_t group_kind, int32_t* __restrict__ ginfo,
                                                       const int32_t* __restrict__ rowptr_b, const int32_t* __restrict__ col_b,
                                                       int32_t* __restrict__ ginfo_b) {
    if (blockIdx.y == 1) { rowptr = rowptr_b; col = col_b; ginfo = ginfo_b; }     // second view of the same rows (gml_csr_group_info2)
    const int group_rows = group_kind == GML_GROUPS64_RANKED ? 64 : group_kind;
    __shared__ int deg[128];
    __shared__ int red[4];
    __shared__ unsigned char row_of_rank[128];
    const int t = threadIdx.x;
    const int64_t g = blockIdx.x;
    const int64_t r0 = g * group_rows;
    const int64_t r1 = min(r0 + group_rows, nrows);
    const int kb = rowptr[r0], ke = rowptr[r1];
    int mn = INT32_MAX, mx = -1;
    for (int k = kb + t; k < ke; k += 128) {
        const int c = col[k];
        mn = min(mn, c);
        mx = max(mx, c);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = min(mn, __shfl_xor(mn, off));
        mx = max(mx, __shfl_xor(mx, off));
    }
    if ((t & 63) == 0) { red[2 * (t >> 6)] = mn; red[2 * (t >> 6) + 1] = mx; }
    if (t < group_rows) deg[t] = (r0 + t < nrows) ? rowptr[r0 + t + 1] - rowptr[r0 + t] : -1;   // rows past the end rank last
    __syncthreads();
    int32_t* rec = ginfo + g * GML_GREC_INTS(group_kind);
    if (t == 0) {
        mn = min(red[0], red[2]);
        mx = max(red[1], red[3]);
        int4 o;
        o.x = kb; o.y = ke - kb; o.z = (ke > kb) ? mn : 0; o.w = (ke > kb) ? mx - mn + 1 : 0;
        *reinterpret_cast<int4*>(rec) = o;
    }
    if (t < group_rows) {
        const int d = deg[t];
        int rank = 0;
        for (int u = 0; u < group_rows; ++u) rank += (deg[u] > d || (deg[u] == d && u < t)) ? 1 : 0;
        row_of_rank[rank] = (unsigned char)t;
    }
    __syncthreads();
    if (t < group_rows) {
        const int wave = t >> 4, i = t & 15;
        if (group_rows == 128) {
            const int a = ((wave & 3) + (int)g) & 3;
            const int blk = (wave < 4) ? a : 7 - a;
            reinterpret_cast<unsigned char*>(rec + 4)[t] = row_of_rank[blk * 16 + i];
        } else if (group_kind == GML_GROUPS64_RANKED) {      // 4 waves, one per SIMD: rank blocks rotate with the group
            reinterpret_cast<unsigned char*>(rec + 4)[t] = row_of_rank[(((wave + (int)g) & 3)) * 16 + i];
        }
    }
}

extern "C" int32_t gml_csr_group_record_ints(int32_t group_rows) {
    return (group_rows == 64 || group_rows == 128 || group_rows == GML_GROUPS64_RANKED) ? GML_GREC_INTS(group_rows) : 0;
}

extern "C" int gml_csr_group_info(const int32_t* rowptr, const int32_t* col, int64_t num_rows, int32_t group_rows,
                                  int32_t* ginfo, gml_stream_t stream) {
    if (num_rows < 0 || (group_rows != 64 && group_rows != 128 && group_rows != GML_GROUPS64_RANKED)) return GML_E_BADARG;
    const int rows = group_rows == GML_GROUPS64_RANKED ? 64 : group_rows;
    if (num_rows == 0) return GML_OK;
    if (!rowptr || !ginfo) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_group_info, dim3((unsigned)gml_cdiv(num_rows, rows)), dim3(128), 0,
                       (hipStream_t)stream, rowptr, col, num_rows, group_rows, ginfo, nullptr, nullptr, nullptr);
    return gml_launch_status();
}

// the group records of BOTH views of one graph structure (target-keyed and source-keyed CSR over the same rows) in one launch
extern "C" int gml_csr_group_info2(const int32_t* rowptr_a, const int32_t* col_a, int32_t* ginfo_a, const int32_t* rowptr_b,
                                   const int32_t* col_b, int32_t* ginfo_b, int64_t num_rows, int32_t group_rows, gml_stream_t stream) {
    if (num_rows < 0 || (group_rows != 64 && group_rows != 128 && group_rows != GML_GROUPS64_RANKED)) return GML_E_BADARG;
    const int rows = group_rows == GML_GROUPS64_RANKED ? 64 : group_rows;
    if (num_rows == 0) return GML_OK;
    if (!rowptr_a || !ginfo_a || !rowptr_b || !ginfo_b) return GML_E_BADARG;
    hipLaunchKernelGGL(gml_k_group_info, dim3((unsigned)gml_cdiv(num_rows, rows), 2), dim3(128), 0,
                       (hipStream_t)stream, rowptr_a, col_a, num_rows, group_rows, ginfo_a, rowptr_b, col_b, ginfo_b);
    return gml_launch_status();
}

// =============================================================================================
// gml_batch_assemble: one launch builds a padded static-shape batch AND its index structure from a device-resident data set
// whose per-graph structure was precomputed once (round 4, VERDICT r03 item 3: the reference's regime is batch 64, shuffled,
// Zinc12k.py:20-22,359 -- the graphs of a data set never change, only which 64 of them form the batch).
// A batch is the block-diagonal union of its graphs in the order given, so every per-batch index array is a per-graph array
// plus the graph's node / edge offset in the batch:
//   source view: the data set keeps each graph's support edges sorted by source (libs/utils.py:608-609): rows of the batch =
//                the graph's rows, rowptr_t = edge offset + local prefix, col_t = local target + node offset, perm_t = identity;
//   target view: per graph the stable target sort of its edges (tperm: k-th target-sorted edge -> its source-order position,
//                tinv its inverse): rowptr = edge offset + local prefix, col = source of that edge + node offset,
//                perm = tpos = edge offset + tperm, pos_t = edge offset + tinv.
// Integer-exact: equal to gml_csr_from_coo / gml_csr_from_sorted_coo / gml_csr_link_transpose on the assembled batch (tested).
// Padding as dataset.DeviceDataset.batch_padded: padding nodes carry zero features and form graph B; padding edges are
// zero-valued self loops dealt dmax per padding node.
// =============================================================================================
__global__ __launch_bounds__(256) void gml_k_batch_assemble(const gml_batch_desc d) {
    extern __shared__ int64_t sh[];                          // nlo[B], elo[B], nnew[B + 1], enew[B + 1]
    int64_t* nlo = sh;
    int64_t* elo = sh + d.B;
    int64_t* nnew = sh + 2 * d.B;
    int64_t* enew = sh + 3 * d.B + 1;
    const int B = d.B;
    for (int g = threadIdx.x; g < B; g += blockDim.x) {
        const int64_t id = d.ids[g];
        const bool has = id >= 0 && id < d.G;
        const int64_t ic = has ? id : 0;
        nlo[g] = d.node_ptr[ic];
        elo[g] = d.edge_ptr2[ic];
        nnew[g + 1] = has ? d.node_ptr[ic + 1] - d.node_ptr[ic] : 0;
        enew[g + 1] = has ? d.edge_ptr2[ic + 1] - d.edge_ptr2[ic] : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        nnew[0] = 0; enew[0] = 0;
        for (int g = 0; g < B; ++g) { nnew[g + 1] += nnew[g]; enew[g + 1] += enew[g]; }
    }
    __syncthreads();
    const int64_t n_real = nnew[B] < d.n_pad ? nnew[B] : d.n_pad, e_real = enew[B] < d.e2_pad ? enew[B] : d.e2_pad;
    auto seg_of = [&](const int64_t* ptr, int64_t i) {       // first g with ptr[g + 1] > i (i below ptr[B])
        int lo = 0, hi = B;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ptr[mid + 1] > i) hi = mid; else lo = mid + 1; }
        return lo;
    };
    // ---- blocks behind the element blocks: the 128-row group records of both CSR views (gml_csr_group_info's records, computed from the
    //      same formulas the element blocks write -- no block reads another block's output), when the caller wants them
    if ((int)blockIdx.x >= d.nblk_main) {
        const int gb = (int)blockIdx.x - d.nblk_main, ngr = (d.n_pad + 127) / 128;
        const int view = gb / ngr;                           // 0: target-keyed (rowptr / col -> ginfo128), 1: source-keyed (-> ginfo_t128)
        const int64_t gidx = gb % ngr;
        auto rowptr_of = [&](int64_t r) -> int {
            if (r >= d.n_pad) return d.e2_pad;
            if (r < n_real) {
                const int g = seg_of(nnew, r);
                const int64_t src = r - nnew[g] + nlo[g];
                return (int)(enew[g] + (view ? d.rp_src[src] : d.rp_dst[src]));
            }
            const int64_t q = e_real + (r - n_real) * (int64_t)d.dmax;
            return (int)(q < d.e2_pad ? q : d.e2_pad);
        };
        auto col_of = [&](int64_t k) -> int {
            if (k < e_real) {
                const int g = seg_of(enew, k);
                const int64_t sp = elo[g] + (k - enew[g]);
                return view ? (int)(d.edge_index2[d.E2all + sp] + nnew[g]) : (int)(d.edge_index2[elo[g] + d.tperm[sp]] + nnew[g]);
            }
            int64_t node = n_real + (k - e_real) / d.dmax;
            return (int)(node > d.n_pad - 1 ? d.n_pad - 1 : node);
        };
        __shared__ int deg[128];
        __shared__ int red[8];
        __shared__ unsigned char row_of_rank[128];
        __shared__ int rpl[129];
        const int t = threadIdx.x;
        const int64_t r0 = gidx * 128, r1 = min(r0 + 128, (int64_t)d.n_pad);
        if (t <= 128) rpl[t] = rowptr_of(min(r0 + t, (int64_t)d.n_pad));
        __syncthreads();
        const int kb = rpl[0], ke = rpl[(int)(r1 - r0)];
        int mn = INT32_MAX, mx = -1;
        for (int k = kb + t; k < ke; k += 256) {
            const int c = col_of(k);
            mn = min(mn, c);
            mx = max(mx, c);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mn = min(mn, __shfl_xor(mn, off));
            mx = max(mx, __shfl_xor(mx, off));
        }
        if ((t & 63) == 0) { red[2 * (t >> 6)] = mn; red[2 * (t >> 6) + 1] = mx; }
        if (t < 128) deg[t] = (r0 + t < d.n_pad) ? rpl[t + 1] - rpl[t] : -1;      // rows past the end rank last
        __syncthreads();
        int32_t* rec = (view ? d.ginfo_t128 : d.ginfo128) + gidx * GML_GREC_INTS(128);
        if (t == 0) {
            mn = min(min(red[0], red[2]), min(red[4], red[6]));
            mx = max(max(red[1], red[3]), max(red[5], red[7]));
            int4 o;
            o.x = kb; o.y = ke - kb; o.z = (ke > kb) ? mn : 0; o.w = (ke > kb) ? mx - mn + 1 : 0;
            *reinterpret_cast<int4*>(rec) = o;
        }
        if (t < 128) {
            const int dg = deg[t];
            int rank = 0;
            for (int u = 0; u < 128; ++u) rank += (deg[u] > dg || (deg[u] == dg && u < t)) ? 1 : 0;
            row_of_rank[rank] = (unsigned char)t;
        }
        __syncthreads();
        if (t < 128) {                                       // (the dealing of rank blocks to waves: gml_k_group_info, 128-row groups)
            const int wave = t >> 4, i16 = t & 15;
            const int a = ((wave & 3) + (int)gidx) & 3;
            const int blk = (wave < 4) ? a : 7 - a;
            reinterpret_cast<unsigned char*>(rec + 4)[t] = row_of_rank[blk * 16 + i16];
        }
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ldo = d.ldx_out > d.F ? d.ldx_out : d.F;
    // ---- graphs
    if (i <= B + 1) {
        d.ptr_out[i] = i <= B ? (int32_t)(nnew[i] < d.n_pad ? nnew[i] : d.n_pad) : d.n_pad;
        if (i < B) {
            const int64_t id = d.ids[i];
            const bool has = id >= 0 && id < d.G;
            d.y_out[i] = has ? d.y[id] : 0.f;
            d.valid_out[i] = has ? 1.f : 0.f;
        } else if (i == B) d.y_out[i] = 0.f;
    }
    // ---- nodes
    if (i <= d.n_pad) {
        if (i == d.n_pad) { d.rowptr[i] = d.e2_pad; d.rowptr_t[i] = d.e2_pad; }
        else if (i < n_real) {
            const int g = seg_of(nnew, i);
            const int64_t src = i - nnew[g] + nlo[g];
            for (int f = 0; f < d.F; ++f) d.x_out[i * ldo + f] = d.x[src * d.F + f];
            for (int f = d.F; f < ldo; ++f) d.x_out[i * ldo + f] = 0.f;
            d.batch_out[i] = g;
            d.rowptr_t[i] = (int32_t)(enew[g] + d.rp_src[src]);
            d.rowptr[i] = (int32_t)(enew[g] + d.rp_dst[src]);
        } else {
            for (int f = 0; f < ldo; ++f) d.x_out[i * ldo + f] = 0.f;
            d.batch_out[i] = B;
            const int64_t r = e_real + (i - n_real) * (int64_t)d.dmax;
            d.rowptr_t[i] = d.rowptr[i] = (int32_t)(r < d.e2_pad ? r : d.e2_pad);
        }
    }
    // ---- support edges
    if (i < d.e2_pad) {
        if (i < e_real) {
            const int g = seg_of(enew, i);
            const int64_t k = i - enew[g], sp = elo[g] + k, base = nnew[g];
            for (int s = 0; s < d.S; ++s) d.ea_out[i * d.S + s] = d.edge_attr2[sp * d.S + s];
            if (d.es) {
                const u32x4* q = reinterpret_cast<const u32x4*>(d.es + sp * 8);
                u32x4* o = reinterpret_cast<u32x4*>(d.es_out + i * 8);
                o[0] = q[0]; o[1] = q[1];
            }
            d.col_t[i] = (int32_t)(d.edge_index2[d.E2all + sp] + base);
            d.pos_t[i] = (int32_t)(enew[g] + d.tinv[sp]);
            const int64_t tp = d.tperm[sp];                  // k-th target-sorted edge of the graph: its source-order position
            d.perm[i] = (int32_t)(enew[g] + tp);
            d.col[i] = (int32_t)(d.edge_index2[elo[g] + tp] + base);
        } else {
            int64_t node = n_real + (i - e_real) / d.dmax;
            if (node > d.n_pad - 1) node = d.n_pad - 1;
            for (int s = 0; s < d.S; ++s) d.ea_out[i * d.S + s] = 0.f;
            if (d.es) {
                u32x4* o = reinterpret_cast<u32x4*>(d.es_out + i * 8);
                o[0] = u32x4{0u, 0u, 0u, 0u}; o[1] = u32x4{0u, 0u, 0u, 0u};
            }
            d.col_t[i] = d.col[i] = (int32_t)node;
            d.pos_t[i] = d.perm[i] = (int32_t)i;
        }
    }
}

extern "C" int gml_batch_assemble(const gml_batch_desc* d, gml_stream_t stream) {
    if (!d || d->B <= 0 || d->B > 4096 || d->n_pad <= 0 || d->e2_pad < 0 || d->dmax <= 0 || d->F <= 0 || d->S <= 0) return GML_E_BADARG;
    if (!d->ids || !d->node_ptr || !d->edge_ptr2 || !d->x || !d->edge_index2 || !d->edge_attr2 || !d->tperm || !d->tinv || !d->rp_src ||
        !d->rp_dst || !d->y || !d->x_out || !d->ea_out || !d->y_out || !d->valid_out || !d->ptr_out || !d->batch_out || !d->rowptr ||
        !d->col || !d->perm || !d->rowptr_t || !d->col_t || !d->pos_t)
        return GML_E_BADARG;
    if (d->es && (((uintptr_t)d->es | (uintptr_t)d->es_out) & 15)) return GML_E_BADARG;
    const int64_t n = (d->n_pad + 1 > d->e2_pad ? d->n_pad + 1 : d->e2_pad);
    const int64_t m = n > d->B + 2 ? n : d->B + 2;
    const size_t lds = (size_t)(4 * d->B + 2) * sizeof(int64_t);
    gml_batch_desc dd = *d;
    if ((dd.ginfo128 == nullptr) != (dd.ginfo_t128 == nullptr)) return GML_E_BADARG;
    dd.nblk_main = (int32_t)gml_cdiv(m, 256);
    const int extra = dd.ginfo128 ? 2 * ((dd.n_pad + 127) / 128) : 0;
    hipLaunchKernelGGL(gml_k_batch_assemble, dim3((unsigned)(dd.nblk_main + extra)), dim3(256), lds, (hipStream_t)stream, dd);
    return gml_launch_status();
}
