// Large-subdomain path of the one-level Schwarz preconditioner: subdomains of up to 1024 dofs.
//
// What it is for: merged block systems -- the monolithic P2/P1 Stokes matrix of Stokes::assemble + BlockMatrix::merge
// (feddlib/problems/specific/Stokes_def.hpp:47-138, feddlib/core/LinearAlgebra/BlockMatrix_def.hpp:119-148), the system
// FROSch's monolithic preconditioner receives under "Preconditioner Method" = "Monolithic"
// (feddlib/problems/Solver/Preconditioner_def.hpp:243-463, stokes/parametersPrec.xml:7-35).  One overlap layer in the
// graph of a P2 matrix is several hundred dofs wide, far beyond the 256 dofs of the register / LDS kernels of
// schwarz.hip, and a graded unstructured mesh puts very different numbers of nodes into the boxes of a regular
// lattice.  So here
//   * the boxes come from a balanced recursive coordinate bisection of the dofs' carrying nodes (host code, like the
//     mesh generators: median splits along the longest edge of the bounding box, deterministic tie break), every box
//     with at most `target` owned dofs (default 120; halved-ish until every overlapping subdomain fits 1024 dofs);
//   * the local matrices are extracted densely, with the velocity dofs ordered before the pressure dofs, and inverted
//     in batches by the blocked Gauss-Jordan sweep of dense.hip on the f64 matrix cores;
//   * the apply streams the needed rows of the inverses exactly like the small path (same slab layout).
// The definition of the operator is the one of schwarz.hip: M^-1 = sum_i P_i A_i^-1 R_i over principal submatrices of
// the Dirichlet-modified matrix, P_i restricted / averaging / full; only the boxes differ.  It can be selected for any
// system (option "schwarz_big" 1) and is the default for merged ones.  The coarse level is not combined with it.
#include "fedd_internal.hpp"
#include <algorithm>
#include <climits>
#include <cmath>
#include <numeric>

namespace fedd {
namespace {

constexpr int NMB = SCHWARZ_NMAX_BIG;

// ---- balanced recursive coordinate bisection (host) ----
// idx[lo, hi) is split at its median along the longest edge of its bounding box until a part holds at most `target`
// points; parts are numbered in left-to-right order of the recursion.  Ties are broken by the point's index, so the
// result depends on the coordinates only.
void rcb(const std::vector<double>& x, int dim, std::vector<int32_t>& idx, int64_t lo, int64_t hi, int64_t target,
         std::vector<int64_t>& cuts) {
    if (hi - lo <= target) {
        cuts.push_back(hi);
        return;
    }
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t k = lo; k < hi; ++k)
        for (int d = 0; d < dim; ++d) {
            const double v = x[(size_t)idx[k] * dim + d];
            mn[d] = std::min(mn[d], v);
            mx[d] = std::max(mx[d], v);
        }
    int ax = 0;
    for (int d = 1; d < dim; ++d)
        if (mx[d] - mn[d] > mx[ax] - mn[ax]) ax = d;
    const int64_t mid = lo + (hi - lo) / 2;
    std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int32_t a, int32_t b) {
        const double va = x[(size_t)a * dim + ax], vb = x[(size_t)b * dim + ax];
        return va < vb || (va == vb && a < b);
    });
    rcb(x, dim, idx, lo, mid, target, cuts);
    rcb(x, dim, idx, mid, hi, target, cuts);
}

// ---- dense extraction: subdomain b of the chunk -> W + slot * stride, velocities before pressures ----
// pos[k] = row / column of local dof k in the dense matrix; rows of ghost dofs (no stored row) are identity rows;
// everything beyond the subdomain's n (up to its 64-block boundary) is identity padding.
__global__ __launch_bounds__(256) void k_big_extract(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_nown,
                                                     const int32_t* __restrict__ sub_dofs,
                                                     const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                     const double* __restrict__ val, int32_t n_stored, int32_t p_off,
                                                     const int32_t* __restrict__ ids, int dstride,
                                                     int first, double* __restrict__ Wall, int64_t ld, int64_t stride,
                                                     int32_t* __restrict__ nblk) {
    __shared__ int32_t sdof[NMB];
    __shared__ int16_t pos[NMB];
    __shared__ int32_t s_nv;
    const int slot = blockIdx.x, tid = threadIdx.x;
    const int b = ids ? ids[first + slot] : first + slot;
    const int n = sub_n[b], no = sub_nown[b];
    double* __restrict__ W = Wall + (int64_t)slot * stride;
    const int np = ((n + 63) / 64) * 64;
    if (tid == 0) {
        nblk[slot] = np / 64;
        s_nv = 0;
    }
    for (int k = tid; k < n; k += 256) sdof[k] = sub_dofs[(int64_t)b * dstride + k];
    __syncthreads();
    // velocities (dof < p_off) keep their relative order in front, pressures follow
    int nv_mine = 0;
    for (int k = tid; k < n; k += 256) nv_mine += sdof[k] < p_off ? 1 : 0;
    atomicAdd(&s_nv, nv_mine);
    for (int64_t e = tid; e < (int64_t)np * np; e += 256) {
        const int r = (int)(e / np), cidx = (int)(e - (int64_t)r * np);
        W[(int64_t)r * ld + cidx] = (r == cidx && r >= n) ? 1.0 : 0.0;
    }
    __syncthreads();
    const int nv = s_nv;
    for (int k = tid; k < n; k += 256) {
        int before_v = 0, before_p = 0;
        const bool isp = sdof[k] >= p_off;
        for (int m = 0; m < k; ++m) {
            const bool mp = sdof[m] >= p_off;
            before_v += mp ? 0 : 1;
            before_p += mp ? 1 : 0;
        }
        pos[k] = (int16_t)(isp ? nv + before_p : before_v);
    }
    __syncthreads();
    // one wave per local row, lanes over the entries of the CSR row
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < n; r += 4) {
        const int32_t g = sdof[r];
        double* __restrict__ Wr = W + (int64_t)pos[r] * ld;
        if (g < n_stored) {
            for (int32_t p = rowptr[g] + lane; p < rowptr[g + 1]; p += 64) {
                const int32_t col = colind[p];
                // owned dofs [0, no) and overlap dofs [no, n) are each sorted
                int lo = 0, hi = no - 1, found = -1;
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int32_t v = sdof[mid];
                    if (v == col) { found = mid; break; }
                    if (v < col) lo = mid + 1;
                    else hi = mid - 1;
                }
                if (found < 0) {
                    lo = no;
                    hi = n - 1;
                    while (lo <= hi) {
                        const int mid = (lo + hi) >> 1;
                        const int32_t v = sdof[mid];
                        if (v == col) { found = mid; break; }
                        if (v < col) lo = mid + 1;
                        else hi = mid - 1;
                    }
                }
                if (found >= 0) Wr[pos[found]] = val[p];
            }
        } else if (lane == 0) {
            Wr[pos[r]] = 1.0;   // ghost row (not stored on this rank): identity
        }
    }
}

// needed rows of the inverse -> slab [c][nrow] (column-major, schwarz.hip's layout), original dof order
__global__ __launch_bounds__(256) void k_big_slab(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_nown,
                                                  const int32_t* __restrict__ sub_dofs, int32_t p_off, int restricted,
                                                  const int32_t* __restrict__ ids, int dstride,
                                                  int first, const double* __restrict__ Wall, int64_t ld, int64_t stride,
                                                  const int64_t* __restrict__ inv_ptr, double* __restrict__ inv) {
    __shared__ int16_t pos[NMB];
    __shared__ int32_t s_nv;
    const int slot = blockIdx.x, tid = threadIdx.x;
    const int b = ids ? ids[first + slot] : first + slot;
    const int n = sub_n[b], no = sub_nown[b];
    const double* __restrict__ W = Wall + (int64_t)slot * stride;
    const int32_t* __restrict__ sd = sub_dofs + (int64_t)b * dstride;
    if (tid == 0) s_nv = 0;
    __syncthreads();
    int nv_mine = 0;
    for (int k = tid; k < n; k += 256) nv_mine += sd[k] < p_off ? 1 : 0;
    atomicAdd(&s_nv, nv_mine);
    __syncthreads();
    const int nv = s_nv;
    for (int k = tid; k < n; k += 256) {
        int before_v = 0, before_p = 0;
        const bool isp = sd[k] >= p_off;
        for (int m = 0; m < k; ++m) {
            const bool mp = sd[m] >= p_off;
            before_v += mp ? 0 : 1;
            before_p += mp ? 1 : 0;
        }
        pos[k] = (int16_t)(isp ? nv + before_p : before_v);
    }
    __syncthreads();
    const int nrow = restricted ? no : n;
    double* __restrict__ slab = inv + inv_ptr[b];
    for (int64_t e = tid; e < (int64_t)n * nrow; e += 256) {
        const int cidx = (int)(e / nrow), r = (int)(e - (int64_t)cidx * nrow);
        slab[e] = W[(int64_t)pos[r] * ld + pos[cidx]];
    }
}

// z (+)= P_i A_i^-1 R_i r: rows in chunks of at most 256, lane (row, s) sums its row over the columns c == s (mod S)
template <bool RESTRICTED>
__global__ __launch_bounds__(256) void k_apply_big(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_nown,
                                                   const int32_t* __restrict__ sub_dofs,
                                                   const int64_t* __restrict__ inv_ptr, const double* __restrict__ inv,
                                                   const double* __restrict__ r, double* __restrict__ z) {
    __shared__ double rsub[NMB];
    __shared__ int32_t sdof[NMB];
    __shared__ double part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = sub_n[b];
    const int nrow = RESTRICTED ? sub_nown[b] : n;
    for (int k = tid; k < n; k += 256) {
        const int32_t d = sub_dofs[(int64_t)b * NMB + k];
        sdof[k] = d;
        rsub[k] = r[d];
    }
    __syncthreads();
    const double* __restrict__ slab = inv + inv_ptr[b];
    const int RC = min(nrow, 256), S = 256 / RC;
    const int rr = tid % RC, s = tid / RC;
    for (int row0 = 0; row0 < nrow; row0 += RC) {
        const int row = row0 + rr;
        double acc = 0.0;
        if (s < S && row < nrow) {
            int cidx = s;
            for (; cidx + 3 * S < n; cidx += 4 * S) {
                const double a0 = __builtin_nontemporal_load(slab + (int64_t)cidx * nrow + row);
                const double a1 = __builtin_nontemporal_load(slab + (int64_t)(cidx + S) * nrow + row);
                const double a2 = __builtin_nontemporal_load(slab + (int64_t)(cidx + 2 * S) * nrow + row);
                const double a3 = __builtin_nontemporal_load(slab + (int64_t)(cidx + 3 * S) * nrow + row);
                acc += a0 * rsub[cidx] + a1 * rsub[cidx + S] + a2 * rsub[cidx + 2 * S] + a3 * rsub[cidx + 3 * S];
            }
            for (; cidx < n; cidx += S) acc += slab[(int64_t)cidx * nrow + row] * rsub[cidx];
        }
        part[tid] = acc;
        __syncthreads();
        if (tid < RC && row0 + tid < nrow) {
            double sum = 0.0;
            for (int q = 0; q < S; ++q) sum += part[q * RC + tid];
            if (RESTRICTED) z[sdof[row0 + tid]] = sum;
            else atomicAdd(&z[sdof[row0 + tid]], sum);
        }
        __syncthreads();
    }
}

__global__ void k_count_mult_big(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_dofs, double* mult) {
    const int b = blockIdx.x;
    const int n = sub_n[b];
    for (int k = threadIdx.x; k < n; k += blockDim.x) atomicAdd(&mult[sub_dofs[(int64_t)b * NMB + k]], 1.0);
}

__global__ void k_div_big(double* __restrict__ z, const double* __restrict__ m, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = z[i] / m[i];
}

__global__ void k_big_flag(const int32_t* __restrict__ sub_n, int32_t nsub, int n_lo, int32_t* __restrict__ flag) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nsub) flag[b] = sub_n[b] > n_lo ? 1 : 0;
}

__global__ void k_big_compact(const int32_t* __restrict__ sub_n, int32_t nsub, int n_lo, const int32_t* __restrict__ pos,
                              int32_t* __restrict__ ids) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nsub && sub_n[b] > n_lo) ids[pos[b]] = b;
}

}  // namespace

// Dense inverses of every subdomain with more than n_lo dofs (lists in d_sub_dofs with `dstride` entries per
// subdomain): extraction, batched Gauss-Jordan on the matrix cores, slab rows; a chunk of subdomains at a time.
int schwarz_dense_batched(fedd_ctx* c, int64_t nsub, int dstride, int n_lo, int32_t p_off, int restricted, int max_n,
                          int32_t* d_bad, const int32_t* d_sub_n_sel) {
    if (!d_sub_n_sel) d_sub_n_sel = c->d_sub_n.p;     // sizes the selection sees (0 = leave this subdomain out)
    FEDD_CHECK(max_n <= NMB && nsub < ((int64_t)1 << 31), "schwarz_dense_batched: subdomain of %d dofs", max_n);
    const dim3 blk(256), gs((unsigned)((nsub + 255) / 256));
    const int32_t* ids = nullptr;
    int64_t nsel = nsub;
    if (n_lo > 0) {
        FEDD_TRY(c->d_big_ids.ensure((size_t)nsub + 1));
        FEDD_TRY(c->d_big_pos.ensure((size_t)nsub + 1));
        hipLaunchKernelGGL(k_big_flag, gs, blk, 0, c->stream, d_sub_n_sel, (int32_t)nsub, n_lo, c->d_big_pos.p);
        FEDD_TRY(exclusive_scan_i32(c, c->d_big_pos.p, c->d_big_pos.p, nsub, &nsel));
        if (nsel == 0) return 0;
        hipLaunchKernelGGL(k_big_compact, gs, blk, 0, c->stream, d_sub_n_sel, (int32_t)nsub, n_lo,
                           (const int32_t*)c->d_big_pos.p, c->d_big_ids.p);
        ids = c->d_big_ids.p;
    }
    const int64_t ld = ((int64_t)max_n + 63) / 64 * 64, stride = ld * ld;
    const int64_t budget = (int64_t)4 << 30;   // bytes of dense workspace
    const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nsel, 65535), budget / (stride * 8)));
    FEDD_TRY(c->d_big_ws.ensure((size_t)chunk * (size_t)stride));
    FEDD_TRY(c->d_big_nblk.ensure((size_t)chunk));
    for (int64_t first = 0; first < nsel; first += chunk) {
        const int nb = (int)std::min<int64_t>(chunk, nsel - first);
        hipLaunchKernelGGL(k_big_extract, dim3(nb), blk, 0, c->stream, (const int32_t*)c->d_sub_n.p,
                           (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p, (const int32_t*)c->d_rowptr.p,
                           (const int32_t*)c->d_colind.p, (const double*)c->d_val.p, (int32_t)c->n_rows_ext, p_off, ids, dstride,
                           (int)first, c->d_big_ws.p, ld, stride, c->d_big_nblk.p);
        FEDD_TRY(dense_invert_batched(c, c->d_big_ws.p, ld, nb, stride, c->d_big_nblk.p, (int)(ld / 64), 0, d_bad));
        hipLaunchKernelGGL(k_big_slab, dim3(nb), blk, 0, c->stream, (const int32_t*)c->d_sub_n.p,
                           (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p, p_off, restricted, ids, dstride,
                           (int)first, (const double*)c->d_big_ws.p, ld, stride, (const int64_t*)c->d_inv_ptr.p, c->d_inv.p);
    }
    FEDD_HIP(hipGetLastError());
    return 0;
}

bool schwarz_use_big(const fedd_ctx* c) { return c->sw_big > 0 || (c->sw_big < 0 && c->merged); }

int schwarz_setup_big(fedd_ctx* c) {
    c->have_coarse = false;
    c->have_schwarz = false;
    FEDD_CHECK(!c->sw_two_level, "schwarz setup: the coarse level is not combined with the large-subdomain path "
                                 "(merged block systems / option schwarz_big)");
    ScopedTimer timer(c, FEDD_T_SCHWARZ_SETUP);
    const int dim = c->dim, dofs = c->dofs;
    const int64_t n_rows = c->n_rows;
    if (c->nranks > 1) {
        // rank-local failures are shared before the first collective of the setup: a rank that returned here on its own
        // would leave the others waiting in the all-reduce below (the small path defers its checks the same way)
        FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>(1, c->d_dtmp0.cap)));
        const double mine = n_rows > 0 ? 0.0 : 1.0;
        FEDD_HIP(hipMemcpyAsync(c->d_dtmp0.p, &mine, sizeof(double), hipMemcpyHostToDevice, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        FEDD_TRY(allreduce_sum(c, c->d_dtmp0.p, 1));
        double any = 0.0;
        FEDD_HIP(hipMemcpyAsync(&any, c->d_dtmp0.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        FEDD_CHECK(any == 0.0, "schwarz setup: %d rank(s) own no rows", (int)any);
    } else {
        FEDD_CHECK(n_rows > 0, "schwarz setup: no owned rows");
    }
    // ---- coordinates of the node that carries each owned dof (host) ----
    std::vector<double> xyz((size_t)c->n_node * dim);
    FEDD_HIP(hipMemcpyAsync(xyz.data(), c->d_xyz.p, xyz.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    std::vector<int32_t> dof_node;
    if (c->merged) {
        dof_node.resize((size_t)n_rows);
        FEDD_HIP(hipMemcpyAsync(dof_node.data(), c->d_dof_node.p, dof_node.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    }
    FEDD_HIP(hipStreamSynchronize(c->stream));
    std::vector<double> x((size_t)n_rows * dim);
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t nd = c->merged ? dof_node[(size_t)r] : r / dofs;
        for (int d = 0; d < dim; ++d) x[(size_t)r * dim + d] = xyz[(size_t)nd * dim + d];
    }
    const int restricted = c->sw_combine == FEDD_COMBINE_RESTRICTED ? 1 : 0;
    int64_t target = c->sw_big_target > 0 ? c->sw_big_target : (c->sw_target > 0 ? (int64_t)c->sw_target * (c->merged ? 1 : dofs) : 120);
    int64_t nsub = 0;
    int32_t max_n = 0, max_own = 0;
    std::vector<int32_t> idx((size_t)n_rows), bin((size_t)n_rows);
    for (int attempt = 0;; ++attempt) {
        // ---- boxes: every rank bisects its own dofs (the ranks' boxes do not have to agree) ----
        std::iota(idx.begin(), idx.end(), 0);
        std::vector<int64_t> cuts;
        rcb(x, dim, idx, 0, n_rows, std::max<int64_t>(1, target), cuts);
        nsub = (int64_t)cuts.size();
        std::vector<int32_t> ptr((size_t)nsub + 1, 0);
        int64_t lo = 0;
        for (int64_t b = 0; b < nsub; ++b) {
            std::sort(idx.begin() + lo, idx.begin() + cuts[(size_t)b]);
            for (int64_t k = lo; k < cuts[(size_t)b]; ++k) bin[(size_t)idx[(size_t)k]] = (int32_t)b;
            ptr[(size_t)b + 1] = (int32_t)cuts[(size_t)b];
            lo = cuts[(size_t)b];
        }
        FEDD_TRY(c->d_bin_ptr.ensure((size_t)nsub + 1));
        FEDD_TRY(c->d_bin_nodes.ensure((size_t)n_rows));
        FEDD_TRY(c->d_node_bin.ensure((size_t)n_rows));
        FEDD_HIP(hipMemcpyAsync(c->d_bin_ptr.p, ptr.data(), ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        FEDD_HIP(hipMemcpyAsync(c->d_bin_nodes.p, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        FEDD_HIP(hipMemcpyAsync(c->d_node_bin.p, bin.data(), bin.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        FEDD_TRY(schwarz_overlap_lists_big(c, nsub, &max_n, &max_own));   // (synchronises: the host vectors may go)
        int32_t max_all = max_n;
        if (c->nranks > 1) {   // every rank takes the same decision (max through the sum transport: own slot per rank)
            std::vector<double> h((size_t)c->nranks, 0.0);
            h[(size_t)c->rank] = (double)max_n;
            FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>((size_t)c->nranks, c->d_dtmp0.cap)));
            FEDD_HIP(hipMemcpyAsync(c->d_dtmp0.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
            FEDD_TRY(allreduce_sum(c, c->d_dtmp0.p, c->nranks));
            FEDD_HIP(hipMemcpyAsync(h.data(), c->d_dtmp0.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            FEDD_HIP(hipStreamSynchronize(c->stream));
            for (double v : h) max_all = std::max(max_all, (int32_t)v);
        }
        c->sw_max_size_all = max_all;
        if (max_all <= NMB || target <= 1 || attempt >= 12) break;
        target = std::max<int64_t>(1, (int64_t)(target * 0.7));
    }
    FEDD_CHECK(c->sw_max_size_all <= NMB, "schwarz setup: an overlapping subdomain has %d dofs, the batched dense solver takes "
               "at most %d (one overlap layer of this matrix alone is larger)", c->sw_max_size_all, NMB);
    c->sw_nsub = nsub;
    c->sw_max_size = max_n;
    c->sw_max_own = max_own;
    FEDD_TRY(schwarz_slab_offsets(c, nsub, restricted));
    // ---- dense inverses, a chunk of subdomains at a time ----
    FEDD_TRY(c->d_flags.ensure(16));
    int32_t* d_bad = c->d_flags.p + 1;
    FEDD_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), c->stream));
    const int32_t p_off = c->merged ? (int32_t)c->merged_nA : INT32_MAX;
    FEDD_TRY(schwarz_dense_batched(c, nsub, NMB, 0, p_off, restricted, max_n, d_bad, nullptr));
    int32_t bad = 0;
    FEDD_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    if (c->nranks > 1) {
        double flag = bad ? 1.0 : 0.0;
        FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>(1, c->d_dtmp0.cap)));
        FEDD_HIP(hipMemcpyAsync(c->d_dtmp0.p, &flag, sizeof(double), hipMemcpyHostToDevice, c->stream));
        FEDD_TRY(allreduce_sum(c, c->d_dtmp0.p, 1));
        FEDD_HIP(hipMemcpyAsync(&flag, c->d_dtmp0.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        bad = flag > 0.0;
    }
    FEDD_CHECK(!bad, "schwarz setup: zero pivot in a local factorisation (matrix singular on a subdomain)");
    if (c->sw_combine == FEDD_COMBINE_AVERAGING) {
        FEDD_TRY(c->d_mult.ensure((size_t)c->n_cols));
        FEDD_HIP(hipMemsetAsync(c->d_mult.p, 0, (size_t)c->n_cols * sizeof(double), c->stream));
        hipLaunchKernelGGL(k_count_mult_big, dim3((unsigned)nsub), dim3(64), 0, c->stream, (const int32_t*)c->d_sub_n.p,
                           (const int32_t*)c->d_sub_dofs.p, c->d_mult.p);
    }
    FEDD_TRY(c->d_ycol.ensure((size_t)c->n_cols));
    FEDD_HIP(hipGetLastError());
    c->have_schwarz = true;
    c->sw_big_active = true;
    timer.stop();
    return 0;
}

int schwarz_apply_big(fedd_ctx* c, const double* d_r_owned, double* d_z_owned, bool r_has_tail) {
    FEDD_CHECK(c->have_schwarz && c->sw_big_active, "schwarz apply: no preconditioner");
    const double* r = d_r_owned;
    if (c->n_cols != c->n_rows || !c->halo.peers.empty()) {   // ghost entries of r
        if (r_has_tail) {
            FEDD_TRY(halo_import(c, const_cast<double*>(d_r_owned), c->dofs));
        } else {
            FEDD_HIP(hipMemcpyAsync(c->d_xcol.p, d_r_owned, (size_t)c->n_rows * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            FEDD_TRY(halo_import(c, c->d_xcol.p, c->dofs));
            r = c->d_xcol.p;
        }
    }
    const dim3 grid((unsigned)c->sw_nsub), blk(256);
    ScopedTimer t(c, FEDD_T_SCHWARZ_APPLY);
    if (c->sw_combine == FEDD_COMBINE_RESTRICTED) {
        hipLaunchKernelGGL(k_apply_big<true>, grid, blk, 0, c->stream, (const int32_t*)c->d_sub_n.p, (const int32_t*)c->d_sub_nown.p,
                           (const int32_t*)c->d_sub_dofs.p, (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r, d_z_owned);
    } else {
        // contributions to ghost dofs are dropped (their rows live on another rank): accumulate on the column
        // vector, copy the owned part
        double* zc = c->d_ycol.p;
        FEDD_HIP(hipMemsetAsync(zc, 0, (size_t)c->n_cols * sizeof(double), c->stream));
        hipLaunchKernelGGL(k_apply_big<false>, grid, blk, 0, c->stream, (const int32_t*)c->d_sub_n.p, (const int32_t*)c->d_sub_nown.p,
                           (const int32_t*)c->d_sub_dofs.p, (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r, zc);
        if (c->sw_combine == FEDD_COMBINE_AVERAGING)
            hipLaunchKernelGGL(k_div_big, dim3((unsigned)((c->n_rows + 255) / 256)), blk, 0, c->stream, zc, (const double*)c->d_mult.p, c->n_rows);
        FEDD_HIP(hipMemcpyAsync(d_z_owned, zc, (size_t)c->n_rows * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    t.stop();
    FEDD_HIP(hipGetLastError());
    return 0;
}

}  // namespace fedd
