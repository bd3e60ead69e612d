// Symbolic phase on the device: node -> (element, local index) adjacency of the owned nodes and
// the CSR pattern of the owned rows.
//
// Replaces what Tpetra discovers dynamically through Matrix::insertGlobalValues + fillComplete
// (feddlib/core/LinearAlgebra/Matrix_def.hpp:88-92,192-199; the inserts are issued at
// feddlib/core/FE/FE_def.hpp:657-659): duplicate (row,col) pairs collapse to one entry, every
// inserted pair stays structurally (also when its value is 0), columns end up sorted.
// No dynamic insert here: counting sort by node, per-row sorted-unique merge in LDS, closed-form
// expansion to dofs.
#include "fedd_internal.hpp"
#include <chrono>

namespace fedd {
namespace {

__global__ void k_count_n2e(const int32_t* __restrict__ conn, int64_t n_ent, int32_t n_own, int32_t* cnt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ent; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t nd = conn[i];
        if (nd < n_own) atomicAdd(&cnt[nd], 1);
    }
}

__global__ void k_fill_n2e(const int32_t* __restrict__ conn, int64_t n_ent, int32_t n_own, int32_t* cursor,
                           int32_t* __restrict__ n2e) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ent; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t nd = conn[i];
        if (nd < n_own) {
            const int32_t pos = atomicAdd(&cursor[nd], 1);
            n2e[pos] = (int32_t)i;
        }
    }
}

// the atomics above leave each node's list in arrival order; sort it so that every later sum over
// the list runs in a fixed (element-id) order => bitwise reproducible assembly.
__global__ void k_sort_n2e(const int32_t* __restrict__ ptr, int32_t n_own, int32_t* n2e) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_own) return;
    const int32_t b = ptr[r], e = ptr[r + 1];
    for (int32_t i = b + 1; i < e; ++i) {
        const int32_t v = n2e[i];
        int32_t j = i - 1;
        while (j >= b && n2e[j] > v) {
            n2e[j + 1] = n2e[j];
            --j;
        }
        n2e[j + 1] = v;
    }
}

// One lane per owned node: merge the node lists of its incident elements into a sorted unique
// list held in LDS (layout [k][lane], so lanes never share a bank row).  FILL = false counts,
// FILL = true writes the columns.
template <bool FILL>
__global__ __launch_bounds__(64) void k_node_pattern(const int32_t* __restrict__ conn, int nen,
                                                     const int32_t* __restrict__ n2e_ptr,
                                                     const int32_t* __restrict__ n2e, int32_t n_own, int cap,
                                                     int32_t* __restrict__ row_cnt,
                                                     const int32_t* __restrict__ rowptr,
                                                     int32_t* __restrict__ colind) {
    extern __shared__ int32_t lst[];
    const int lane = threadIdx.x;
    const int32_t r = blockIdx.x * 64 + lane;
    if (r >= n_own) return;
    int len = 0;
    int32_t seen0 = -1, seen1 = -1;     // the two most recent candidates: neighbouring elements repeat their nodes
    auto insert = [&](int32_t col) {
        if (col == seen0 || col == seen1) return;
        seen1 = seen0;
        seen0 = col;
        // lower bound by bisection (lists of ~15-30 entries: 4-5 dependent LDS reads instead of ~len / 2)
        int lo = 0, hi = len;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (lst[mid * 64 + lane] < col) lo = mid + 1;
            else hi = mid;
        }
        const int pos = lo;
        if (pos < len && lst[pos * 64 + lane] == col) return;
        if (len < cap) {
            for (int k = len; k > pos; --k) lst[k * 64 + lane] = lst[(k - 1) * 64 + lane];
            lst[pos * 64 + lane] = col;
            ++len;
        }
    };
    // The incident elements are taken eight at a time: their ids, then their first four node ids, are
    // loaded as independent requests (two memory latencies per batch instead of two per element);
    // nodes beyond the fourth (P2) follow one by one.  Insertion order is unchanged.
    const int32_t pb = n2e_ptr[r], pe = n2e_ptr[r + 1];
    for (int32_t p0 = pb; p0 < pe; p0 += 8) {
        int32_t ee[8], c4[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) ee[u] = p0 + u < pe ? n2e[p0 + u] / nen : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) c4[u][j] = (ee[u] >= 0 && j < nen) ? conn[(int64_t)ee[u] * nen + j] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (ee[u] < 0) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c4[u][j] >= 0) insert(c4[u][j]);
            for (int j = 4; j < nen; ++j) insert(conn[(int64_t)ee[u] * nen + j]);
        }
    }
    if (!FILL) {
        row_cnt[r] = len;
        // stash the merged list ([k][node] layout: coalesced) so that the fill pass is a copy
        if (colind)
            for (int k = 0; k < len; ++k) colind[(int64_t)k * n_own + r] = lst[k * 64 + lane];
    } else {
        const int32_t b = rowptr[r];
        for (int k = 0; k < len; ++k) colind[b + k] = lst[k * 64 + lane];
    }
}

// The same merge for vertex-only elements (lists of at most NP_CAP = 32 nodes), without the ordered insertion: in lockstep a
// wave pays every lane's shifting at nearly every candidate (2.48 ms for 9.9 M nodes).  Here a lane appends new nodes to its
// list in LDS and finds repeats through a 64-slot open-addressing table of list positions (one byte each); the finished list
// is sorted in registers by a bitonic network and written to the stash plane by plane (coalesced).  Same lists, same order.
// (NEN = 4: an element's nodes are one 16-byte load -- with a lane per node every load instruction of this kernel touches 64
// cache lines, and the lines looked up, 120 per node, are what bounded it)
constexpr int NP_CAP = 32, NP_T = 64;
template <int NEN>
__global__ __launch_bounds__(64) void k_node_pattern_hash(const int32_t* __restrict__ conn, const int32_t* __restrict__ n2e_ptr,
                                                          const int32_t* __restrict__ n2e, int32_t n_own, int32_t* __restrict__ row_cnt,
                                                          int32_t* __restrict__ stash) {
    __shared__ int32_t lst[NP_CAP][64];
    __shared__ uint8_t tab[NP_T][64];
    const int lane = threadIdx.x;
    const int32_t r = blockIdx.x * 64 + lane;
    if (r >= n_own) return;
#pragma unroll
    for (int h = 0; h < NP_T; ++h) tab[h][lane] = 0xFF;
    int len = 0;
    bool over = false;
    int32_t seen0 = -1, seen1 = -1;
    auto insert = [&](int32_t col) {
        if (col == seen0 || col == seen1) return;
        seen1 = seen0;
        seen0 = col;
        uint32_t h = ((uint32_t)col * 0x9E3779B1u) >> 26;
        for (;;) {
            const uint8_t e = tab[h][lane];
            if (e == 0xFF) {
                if (len < NP_CAP) {
                    tab[h][lane] = (uint8_t)len;
                    lst[len][lane] = col;
                    ++len;
                } else {
                    over = true;
                }
                return;
            }
            if (lst[e][lane] == col) return;
            h = (h + 1) & (NP_T - 1);
        }
    };
    const int32_t pb = n2e_ptr[r], pe = n2e_ptr[r + 1];
    for (int32_t p0 = pb; p0 < pe; p0 += 8) {
        int32_t ee[8], c4[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) ee[u] = p0 + u < pe ? n2e[p0 + u] / NEN : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (NEN == 4) {
                const int4 q = ee[u] >= 0 ? reinterpret_cast<const int4*>(conn)[ee[u]] : make_int4(-1, -1, -1, -1);
                c4[u][0] = q.x; c4[u][1] = q.y; c4[u][2] = q.z; c4[u][3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) c4[u][j] = (ee[u] >= 0 && j < NEN) ? conn[(int64_t)ee[u] * NEN + j] : -1;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (ee[u] < 0) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c4[u][j] >= 0) insert(c4[u][j]);
        }
    }
    int32_t v[NP_CAP];
#pragma unroll
    for (int k = 0; k < NP_CAP; ++k) v[k] = k < len ? lst[k][lane] : 0x7FFFFFFF;
#pragma unroll
    for (int k = 2; k <= NP_CAP; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < NP_CAP; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const int32_t a = v[i], b = v[l];
                    const bool up = (i & k) == 0;
                    v[i] = up ? min(a, b) : max(a, b);
                    v[l] = up ? max(a, b) : min(a, b);
                }
            }
    row_cnt[r] = over ? NP_CAP : len;     // (a full list sends the caller to the general kernel with the full bound)
#pragma unroll
    for (int k = 0; k < NP_CAP; ++k)
        if (k < len) stash[(int64_t)k * n_own + r] = v[k];
}

// fill pass when the count pass stashed its lists: stash[k][node] -> colind[rowptr[node] + k]
// A workgroup takes 128 rows, i.e. one contiguous run of colind: the lists are read coalesced ([k][node] layout), put in
// place in LDS and written out as one coalesced stream (a lane writing its own row's entries made every store instruction
// touch 64 cache lines: 1.33 ms for 148 M entries).  cap = longest list the stash holds.
__global__ __launch_bounds__(128) void k_pattern_compact(const int32_t* __restrict__ stash, const int32_t* __restrict__ rowptr,
                                                         int32_t n_own, int cap, int32_t* __restrict__ colind) {
    extern __shared__ int32_t sh[];         // [128 * cap]
    const int tid = threadIdx.x;
    const int32_t R0 = blockIdx.x * 128, R1 = min(n_own, R0 + 128);
    const int32_t b0 = rowptr[R0], total = rowptr[R1] - b0;
    const int32_t r = R0 + tid;
    if (r < R1) {
        const int32_t b = rowptr[r] - b0, len = rowptr[r + 1] - rowptr[r];
        // (eight planes of the stash at a time: the loads in flight together, then the stores to LDS)
        for (int k0 = 0; k0 < len; k0 += 8) {
            int32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k0 + u < len ? stash[(int64_t)(k0 + u) * n_own + r] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + u < len) sh[b + k0 + u] = v[u];
        }
    }
    __syncthreads();
    for (int32_t i = tid; i < total; i += 128) colind[b0 + i] = sh[i];
}

// node pattern -> dof pattern, closed form (no scan): node-wise interleaved dofs
// (feddlib/core/LinearAlgebra/Map_def.hpp:101-104).
__global__ void k_expand_pattern(const int32_t* __restrict__ nptr, const int32_t* __restrict__ ncol, int32_t n_own,
                                 int dofs, int full, int32_t* __restrict__ rowptr, int32_t* __restrict__ colind) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_rows = (int64_t)n_own * dofs;
    if (row > n_rows) return;
    if (row == n_rows) {
        rowptr[row] = nptr[n_own] * dofs * (full ? dofs : 1);
        return;
    }
    const int32_t nd = (int32_t)(row / dofs);
    const int a = (int)(row % dofs);
    const int32_t nb = nptr[nd], nn = nptr[nd + 1] - nb;
    if (full) {
        const int32_t start = nb * dofs * dofs + a * nn * dofs;
        rowptr[row] = start;
        for (int32_t s = 0; s < nn; ++s)
            for (int b = 0; b < dofs; ++b) colind[start + s * dofs + b] = ncol[nb + s] * dofs + b;
    } else {
        const int32_t start = nb * dofs + a * nn;
        rowptr[row] = start;
        for (int32_t s = 0; s < nn; ++s) colind[start + s] = ncol[nb + s] * dofs + a;
    }
}

}  // namespace

static int build_adjacency_impl(fedd_ctx* c);
int build_adjacency(fedd_ctx* c) {      // once per mesh; its wall time is kept for fedd_mesh_setup_info
    FEDD_HIP(hipStreamSynchronize(c->stream));
    const auto t0 = std::chrono::steady_clock::now();
    FEDD_TRY(build_adjacency_impl(c));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    c->adj_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

static int build_adjacency_impl(fedd_ctx* c) {
    const int64_t n_ent = c->n_elem * c->nen;
    const int32_t n_own = (int32_t)(c->n_own + c->n_rowg);   // every node that gets rows
    FEDD_TRY(c->d_n2e_ptr.ensure((size_t)n_own + 1));
    FEDD_TRY(c->d_itmp0.ensure((size_t)n_own + 1));
    int32_t* cnt = c->d_itmp0.p;
    FEDD_HIP(hipMemsetAsync(cnt, 0, ((size_t)n_own + 1) * sizeof(int32_t), c->stream));
    const int nb = (int)std::min<int64_t>(4096, std::max<int64_t>(1, (n_ent + 255) / 256));
    if (n_ent > 0) hipLaunchKernelGGL(k_count_n2e, dim3(nb), dim3(256), 0, c->stream, c->d_conn.p, n_ent, n_own, cnt);
    int32_t md = 0;
    FEDD_TRY(reduce_max_i32(c, cnt, n_own, &md));
    c->max_deg = md;
    int64_t total = 0;
    FEDD_TRY(exclusive_scan_i32(c, cnt, c->d_n2e_ptr.p, n_own, &total));
    FEDD_TRY(c->d_n2e.ensure((size_t)total));
    FEDD_HIP(hipMemcpyAsync(cnt, c->d_n2e_ptr.p, (size_t)n_own * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    if (n_ent > 0) {
        hipLaunchKernelGGL(k_fill_n2e, dim3(nb), dim3(256), 0, c->stream, c->d_conn.p, n_ent, n_own, cnt, c->d_n2e.p);
        hipLaunchKernelGGL(k_sort_n2e, dim3((n_own + 255) / 256), dim3(256), 0, c->stream, c->d_n2e_ptr.p, n_own, c->d_n2e.p);
    }
    FEDD_HIP(hipGetLastError());
    c->have_adj = true;
    return 0;
}

int build_pattern(fedd_ctx* c, int dofs, int block_mode) {
    c->cs_valid = false;   // the solver's compacted SpMV stream follows the matrix values
    const int32_t n_own = (int32_t)(c->n_own + c->n_rowg);   // every node that gets rows (owned, then row ghosts)
    const int nen = c->nen;
    // upper bound for the distinct columns of one node row
    int cap_full = c->max_deg * (nen - 1) + 1;
    if (cap_full < 1) cap_full = 1;
    // The per-lane lists live in LDS (cap x 64 ints per one-wave workgroup), and LDS decides how many waves a CU
    // holds: the bound above (73 for P1 tets) gives 8, a list of 32 entries 20.  Vertex-only elements are tried
    // with 32 first; a row that fills it (then the count may be cut off) sends the pass again with the full bound.
    int cap = (nen == c->dim + 1) ? std::min(cap_full, 32) : cap_full;
  retry_with_full_lists:
    const size_t lds = (size_t)cap * 64 * sizeof(int32_t);
    FEDD_CHECK(lds <= 160 * 1024, "pattern build: a node with %d incident elements exceeds the LDS list (cap %d)", c->max_deg, cap);
    const dim3 grid((n_own + 63) / 64), block(64);
    const bool scalar = dofs == 1;
    // node-level pattern goes straight into the final arrays when dofs == 1
    FEDD_TRY(c->d_itmp1.ensure((size_t)n_own + 1));  // node row counts -> node rowptr
    int32_t* nptr = c->d_itmp1.p;
    if (lds > 64 * 1024)
        FEDD_HIP(hipFuncSetAttribute((const void*)k_node_pattern<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // the count pass stashes its merged lists (cap x n_own ints, <= 8 GiB of the 288; only the first
    // max-row-length planes of it are ever touched) so the fill pass is a copy;
    // beyond that size the fill pass merges again
    int32_t* stash = nullptr;
    if ((int64_t)cap * n_own <= ((int64_t)1 << 31)) {
        FEDD_TRY(c->d_pat_stash.ensure((size_t)cap * (size_t)n_own));
        stash = c->d_pat_stash.p;
    }
    if (c->pat_hash && cap <= NP_CAP && nen == c->dim + 1 && (nen == 3 || nen == 4) && stash)
        if (nen == 4) hipLaunchKernelGGL(k_node_pattern_hash<4>, grid, block, 0, c->stream, c->d_conn.p, c->d_n2e_ptr.p, c->d_n2e.p, n_own, nptr, stash);
        else hipLaunchKernelGGL(k_node_pattern_hash<3>, grid, block, 0, c->stream, c->d_conn.p, c->d_n2e_ptr.p, c->d_n2e.p, n_own, nptr, stash);
    else
        hipLaunchKernelGGL(k_node_pattern<false>, grid, block, lds, c->stream, c->d_conn.p, nen, c->d_n2e_ptr.p,
                           c->d_n2e.p, n_own, cap, nptr, (const int32_t*)nullptr, stash);
    int32_t max_nn = 0;
    FEDD_TRY(reduce_max_i32(c, nptr, n_own, &max_nn));
    if (max_nn >= cap && cap < cap_full) {
        cap = cap_full;
        goto retry_with_full_lists;
    }
    int64_t node_nnz = 0;
    FEDD_TRY(exclusive_scan_i32(c, nptr, nptr, n_own, &node_nnz));
    const int64_t mult = scalar ? 1 : (block_mode == FEDD_BLOCK_FULL ? (int64_t)dofs * dofs : dofs);
    const int64_t nnz = node_nnz * mult;
    FEDD_CHECK(nnz < ((int64_t)1 << 31), "pattern build: %lld nonzeros exceed 32-bit local offsets", (long long)nnz);
    c->dofs = dofs;
    c->block_mode = block_mode;
    c->n_rows = c->n_own * dofs;
    c->n_rows_ext = (int64_t)n_own * dofs;
    c->n_cols = c->n_node * dofs;
    c->nnz_ext = nnz;
    c->nnz = nnz;
    if (c->n_rowg > 0) {   // nonzeros of the owned rows: the node row pointer at the first row ghost
        int32_t owned_node_nnz = 0;
        FEDD_HIP(hipMemcpyAsync(&owned_node_nnz, nptr + c->n_own, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        c->nnz = (int64_t)owned_node_nnz * mult;
    }
    c->max_row_nnz = max_nn * (block_mode == FEDD_BLOCK_FULL ? dofs : 1);
    FEDD_TRY(c->d_rowptr.ensure((size_t)c->n_rows_ext + 1));
    FEDD_TRY(c->d_colind.ensure((size_t)nnz));
    FEDD_TRY(c->d_val.ensure((size_t)nnz));
    FEDD_TRY(c->d_rhs.ensure((size_t)c->n_rows_ext));      // the tail past n_rows is scratch of the Dirichlet kernels
    FEDD_TRY(c->d_x.ensure((size_t)c->n_rows));
    FEDD_TRY(c->d_xcol.ensure((size_t)c->n_cols));
    FEDD_TRY(c->d_isdir.ensure((size_t)c->n_rows_ext));
    FEDD_HIP(hipMemsetAsync(c->d_val.p, 0, (size_t)nnz * sizeof(double), c->stream));
    FEDD_HIP(hipMemsetAsync(c->d_rhs.p, 0, (size_t)c->n_rows_ext * sizeof(double), c->stream));
    FEDD_HIP(hipMemsetAsync(c->d_x.p, 0, (size_t)c->n_rows * sizeof(double), c->stream));
    FEDD_HIP(hipMemsetAsync(c->d_xcol.p, 0, (size_t)c->n_cols * sizeof(double), c->stream));
    FEDD_HIP(hipMemsetAsync(c->d_isdir.p, 0, (size_t)c->n_rows_ext * sizeof(int32_t), c->stream));
    int32_t* ncol = c->d_colind.p;
    if (!scalar) {
        FEDD_TRY(c->d_itmp2.ensure((size_t)node_nnz));
        ncol = c->d_itmp2.p;
    }
    if (stash) {
        const size_t lds_c = (size_t)128 * (size_t)std::max(1, max_nn) * sizeof(int32_t);     // 128 rows of at most max_nn entries
        if (lds_c > 64 * 1024)
            FEDD_HIP(hipFuncSetAttribute((const void*)k_pattern_compact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
        hipLaunchKernelGGL(k_pattern_compact, dim3((unsigned)((n_own + 127) / 128)), dim3(128), lds_c, c->stream,
                           (const int32_t*)stash, (const int32_t*)nptr, n_own, max_nn, ncol);
    } else {
        if (lds > 64 * 1024)
            FEDD_HIP(hipFuncSetAttribute((const void*)k_node_pattern<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_node_pattern<true>, grid, block, lds, c->stream, c->d_conn.p, nen, c->d_n2e_ptr.p,
                           c->d_n2e.p, n_own, cap, (int32_t*)nullptr, (const int32_t*)nptr, ncol);
    }
    if (scalar) {
        FEDD_HIP(hipMemcpyAsync(c->d_rowptr.p, nptr, ((size_t)n_own + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    } else {
        const int64_t nthreads = c->n_rows_ext + 1;
        hipLaunchKernelGGL(k_expand_pattern, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream,
                           (const int32_t*)nptr, (const int32_t*)ncol, n_own, dofs,
                           block_mode == FEDD_BLOCK_FULL ? 1 : 0, c->d_rowptr.p, c->d_colind.p);
    }
    FEDD_HIP(hipGetLastError());
    c->have_pattern = true;
    c->have_schwarz = false;
    c->spmv_rows_ready = false;
    c->merged = false;
    return 0;
}

}  // namespace fedd
