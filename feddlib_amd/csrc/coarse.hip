// Coarse level of the two-level Schwarz operator:  z += Phi K0^-1 Phi^T r.
//
// Stands in for the coarse operator the reference gets from FROSch under "TwoLevel" = true
//   feddlib/problems/tests/laplace/parametersPrec.xml:13, 62-122 (GDSWCoarseOperator, IPOUHarmonic)
//   feddlib/problems/Solver/Preconditioner_def.hpp:243-463 (where the list is handed to Thyra).
// GDSW builds its space from the interface of a few large subdomains (one per rank); this library
// runs thousands of small subdomains per GPU, so the coarse space is geometric instead (normative
// definition, DESIGN.md "two-level"; the tests check it against an independent CPU restatement):
//   Phi   = multilinear hat functions of a regular lattice of (g_d + 1) points per direction over
//           the global bounding box, evaluated at the node that carries the dof, one copy per dof
//           component; rows of Dirichlet dofs are zero,
//   K0    = Phi^T A Phi (dense, lattice dofs without support get a unit diagonal, the others a
//           relative diagonal shift of 1e-12), inverted explicitly and replicated on every rank.
// Setup: nodes are grouped by lattice cell with a stable radix split (deterministic order), one
// wave per cell forms the cell's 2^dim x 4^dim Galerkin block, a gather kernel sums the blocks of
// the <= 2^dim cells around each lattice point in a fixed order, and K0 is inverted in place by a
// blocked Gauss-Jordan sweep whose rank-64 updates run on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64) -- the one genuinely dense contraction of the solver.
// Apply: cell-wise restriction (fixed summation order), dense K0^-1 r0, prolongation.
#include "fedd_internal.hpp"
#include <algorithm>
#include <cmath>

namespace fedd {
namespace {

constexpr int NB = 64;  // block size of the dense inversion

struct Loc {
    int i0[3];
    double f[3];
};

template <int DIM>
__device__ __forceinline__ Loc locate(const CoarseGeom& cg, const double* __restrict__ xyz, int32_t node) {
    Loc o;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        o.i0[d] = 0;
        o.f[d] = 0.0;
    }
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        const double t = (xyz[(int64_t)node * DIM + d] - cg.lo[d]) / cg.L[d] * (double)cg.g[d];
        int i = (int)floor(t);
        i = min(cg.g[d] - 1, max(0, i));
        o.i0[d] = i;
        o.f[d] = t - (double)i;
    }
    return o;
}

template <int DIM>
__device__ __forceinline__ double corner_weight(const Loc& o, int a) {
    double w = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) w *= ((a >> d) & 1) ? o.f[d] : 1.0 - o.f[d];
    return w;
}

template <int DIM>
__device__ __forceinline__ int32_t lattice_node(const CoarseGeom& cg, const int i0[3], int a) {
    int32_t id = 0, mul = 1;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        id += mul * (i0[d] + ((a >> d) & 1));
        mul *= cg.np[d];
    }
    return id;
}

template <int DIM>
__device__ __forceinline__ int32_t cell_of(const CoarseGeom& cg, const int i0[3]) {
    int32_t id = 0, mul = 1;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        id += mul * i0[d];
        mul *= cg.g[d];
    }
    return id;
}

// ---- nodes -> lattice cells ----
template <int DIM>
__global__ void k_cell_key(CoarseGeom cg, const double* __restrict__ xyz, int32_t n, int32_t* __restrict__ key,
                           int32_t* __restrict__ val) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Loc o = locate<DIM>(cg, xyz, i);
    key[i] = cell_of<DIM>(cg, o.i0);
    val[i] = i;
}

// cell_ptr from the sorted keys: position i opens every cell in (key[i-1], key[i]]
__global__ void k_cell_bounds(const int32_t* __restrict__ key, int32_t n, int32_t ncell, int32_t* __restrict__ ptr) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t k = key[i], prev = i > 0 ? key[i - 1] : -1;
    for (int32_t c = prev + 1; c <= k; ++c) ptr[c] = i;
    if (i == n - 1)
        for (int32_t c = k + 1; c <= ncell; ++c) ptr[c] = n;
}

// one stable radix-split pass on bit `bit`: zeros first, both halves keep their order
__global__ void k_split_flags(const int32_t* __restrict__ key, int32_t n, int bit, int32_t* __restrict__ flag) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = ((key[i] >> bit) & 1) ? 0 : 1;
}

__global__ void k_split_scatter(const int32_t* __restrict__ key, const int32_t* __restrict__ val,
                                const int32_t* __restrict__ pos, int32_t n, int bit, int32_t* __restrict__ key_out,
                                int32_t* __restrict__ val_out) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t zeros = pos[n - 1] + (((key[n - 1] >> bit) & 1) ? 0 : 1);
    const int32_t k = key[i];
    const int32_t dst = ((k >> bit) & 1) ? zeros + (i - pos[i]) : pos[i];
    key_out[dst] = k;
    val_out[dst] = val[i];
}

// mask = 1 on free dofs, 0 on Dirichlet dofs; n_free += number of free dofs (integer; one atomic per
// workgroup: one per wave on a single address took 180 us for a million dofs)
__global__ __launch_bounds__(256) void k_mask(const int32_t* __restrict__ isdir, int64_t n, double* __restrict__ mask,
                                              int32_t* n_free) {
    __shared__ int32_t cnt[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool free_dof = i < n && !isdir[i];
    if (i < n) mask[i] = free_dof ? 1.0 : 0.0;
    const unsigned long long b = __ballot(free_dof);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = (int32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int32_t t = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        if (t) atomicAdd(n_free, t);
    }
}

// ---- per-cell Galerkin block ----
// One wave per lattice cell.  For the nodes i of the cell (64 at a time, one per lane) and the
// component pair (k, l): P[slot][lane] = sum_j A_(i,k),(j,l) * w_j(slot) over the 4^DIM lattice
// points around the cell, then out[a][slot] += sum_lane w_i(a) * P[slot][lane] with lane = slot.
// Every sum runs in a fixed order: the result does not depend on scheduling.
template <int DIM>
__global__ __launch_bounds__(64) void k_cell_galerkin(CoarseGeom cg, const int32_t* __restrict__ cell_ptr,
                                                      const int32_t* __restrict__ cell_nodes,
                                                      const double* __restrict__ xyz,
                                                      const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ colind,
                                                      const double* __restrict__ val,
                                                      const double* __restrict__ mask, int dofs,
                                                      double* __restrict__ cellK, int32_t* __restrict__ bad) {
    // grid = (cells, dofs * dofs component pairs, chunks): chunk z takes the 64-node batches
    // z, z + gridDim.z, ... of the cell; the gather kernel adds the chunks in order
    constexpr int NC = 1 << DIM, NS = DIM == 3 ? 64 : 16;
    __shared__ double P[NS][65];
    __shared__ double W[NC][64];
    const int cell = blockIdx.x, lane = threadIdx.x;
    const int k = blockIdx.y / dofs, l = blockIdx.y - k * dofs;
    const int nch = gridDim.z, ch = blockIdx.z;
    int cc[3] = {0, 0, 0};
    {
        int r = cell;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            cc[d] = r % cg.g[d];
            r /= cg.g[d];
        }
    }
    const int32_t nb = cell_ptr[cell], ne = cell_ptr[cell + 1];
        {
            double acc[NC];
#pragma unroll
            for (int a = 0; a < NC; ++a) acc[a] = 0.0;
            for (int32_t base = nb + 64 * ch; base < ne; base += 64 * nch) {
#pragma unroll 8
                for (int s = 0; s < NS; ++s) P[s][lane] = 0.0;
                const int32_t node = base + lane < ne ? cell_nodes[base + lane] : -1;
                if (node >= 0) {
                    const Loc oi = locate<DIM>(cg, xyz, node);
                    const int32_t row = node * dofs + k;
                    const double mrow = mask[row];
#pragma unroll
                    for (int a = 0; a < NC; ++a) W[a][lane] = corner_weight<DIM>(oi, a) * mrow;
                    if (mrow != 0.0) {
                        for (int32_t p = rowptr[row]; p < rowptr[row + 1]; ++p) {
                            const int32_t col = colind[p];
                            const int32_t jn = col / dofs;
                            if (col - jn * dofs != l) continue;
                            const double v = val[p] * mask[col];
                            if (v == 0.0) continue;
                            const Loc oj = locate<DIM>(cg, xyz, jn);
                            int rel[3] = {0, 0, 0};
                            bool ok = true;
#pragma unroll
                            for (int d = 0; d < DIM; ++d) {
                                rel[d] = oj.i0[d] - cc[d] + 1;
                                ok = ok && rel[d] >= 0 && rel[d] <= 2;
                            }
                            if (!ok) {
                                bad[0] = 1;  // a matrix entry couples cells that are not neighbours
                                continue;
                            }
#pragma unroll
                            for (int b = 0; b < NC; ++b) {
                                int slot = 0;
#pragma unroll
                                for (int d = DIM - 1; d >= 0; --d) slot = slot * 4 + rel[d] + ((b >> d) & 1);
                                P[slot][lane] += v * corner_weight<DIM>(oj, b);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < NC; ++a) W[a][lane] = 0.0;
                }
                __syncthreads();
                if (lane < NS) {
                    for (int t = 0; t < 64; ++t) {
                        const double pv = P[lane][t];
#pragma unroll
                        for (int a = 0; a < NC; ++a) acc[a] = fma(W[a][t], pv, acc[a]);
                    }
                }
                __syncthreads();
            }
            if (lane < NS) {
#pragma unroll
                for (int a = 0; a < NC; ++a)
                    cellK[(((((int64_t)cell * nch + ch) * dofs + k) * dofs + l) * NC + a) * NS + lane] = acc[a];
            }
        }
}

// K0[(I,k)][(J,l)] = sum over the cells around lattice point I of their block entry for J
template <int DIM>
__global__ void k_coarse_gather(CoarseGeom cg, int dofs, int nch, int64_t n_lat, const double* __restrict__ cellK,
                                double* __restrict__ K, int64_t ld) {
    constexpr int NC = 1 << DIM, NS = DIM == 3 ? 64 : 16, NW = DIM == 3 ? 125 : 25;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lat * NW) return;
    const int64_t I = t / NW;
    int w = (int)(t - I * NW);
    int iI[3] = {0, 0, 0}, iJ[3] = {0, 0, 0};
    {
        int64_t r = I;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            iI[d] = (int)(r % cg.np[d]);
            r /= cg.np[d];
            iJ[d] = iI[d] + (w % 5) - 2;
            w /= 5;
            if (iJ[d] < 0 || iJ[d] >= cg.np[d]) return;
        }
    }
    int64_t J = 0;
    {
        int64_t mul = 1;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            J += mul * iJ[d];
            mul *= cg.np[d];
        }
    }
    for (int k = 0; k < dofs; ++k)
        for (int l = 0; l < dofs; ++l) {
            double sum = 0.0;
            for (int a = 0; a < NC; ++a) {
                int cell = 0, mul = 1, slot = 0, smul = 1;
                bool ok = true;
#pragma unroll
                for (int d = 0; d < DIM; ++d) {
                    const int cd = iI[d] - ((a >> d) & 1);
                    const int sd = iJ[d] - cd + 1;
                    ok = ok && cd >= 0 && cd < cg.g[d] && sd >= 0 && sd <= 3;
                    cell += mul * cd;
                    mul *= cg.g[d];
                    slot += smul * sd;
                    smul *= 4;
                }
                if (ok)
                    for (int ch = 0; ch < nch; ++ch)
                        sum += cellK[(((((int64_t)cell * nch + ch) * dofs + k) * dofs + l) * NC + a) * NS + slot];
            }
            K[(I * dofs + k) * ld + J * dofs + l] = sum;
        }
}

// rows without any entry (lattice dofs without support) and the padding rows get a unit diagonal,
// the others the relative diagonal shift; one wave per row
__global__ __launch_bounds__(256) void k_fix_diag(double* __restrict__ K, int64_t ld, int64_t n0) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= ld) return;
    if (row >= n0) {
        if (lane == 0) K[row * ld + row] = 1.0;
        return;
    }
    bool any = false;
    for (int64_t j = lane; j < n0; j += 64) any = any || K[row * ld + j] != 0.0;
    const bool row_any = __ballot(any) != 0ull;
    if (lane == 0) K[row * ld + row] = row_any ? K[row * ld + row] * (1.0 + 1e-12) : 1.0;
}

// (the blocked Gauss-Jordan inversion of K0 on the f64 matrix cores lives in dense.hip: dense_invert_batched)

// ---- apply ----
template <int DIM, int DOFS>
__global__ __launch_bounds__(256) void k_restrict_cells(CoarseGeom cg, const int32_t* __restrict__ cell_ptr,
                                                        const int32_t* __restrict__ cell_nodes,
                                                        const double* __restrict__ xyz,
                                                        const double* __restrict__ mask,
                                                        const double* __restrict__ r, double* __restrict__ part) {
    constexpr int NC = 1 << DIM, NV = NC * DOFS;
    __shared__ double red[4][NV];
    const int cell = blockIdx.x, tid = threadIdx.x;
    double acc[NC][DOFS];
#pragma unroll
    for (int a = 0; a < NC; ++a)
#pragma unroll
        for (int k = 0; k < DOFS; ++k) acc[a][k] = 0.0;
    for (int32_t idx = cell_ptr[cell] + tid; idx < cell_ptr[cell + 1]; idx += 256) {
        const int32_t node = cell_nodes[idx];
        const Loc o = locate<DIM>(cg, xyz, node);
        double rv[DOFS];
#pragma unroll
        for (int k = 0; k < DOFS; ++k) rv[k] = r[(int64_t)node * DOFS + k] * mask[(int64_t)node * DOFS + k];
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            const double w = corner_weight<DIM>(o, a);
#pragma unroll
            for (int k = 0; k < DOFS; ++k) acc[a][k] = fma(w, rv[k], acc[a][k]);
        }
    }
#pragma unroll
    for (int a = 0; a < NC; ++a)
#pragma unroll
        for (int k = 0; k < DOFS; ++k) {
            double v = acc[a][k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((tid & 63) == 0) red[tid >> 6][a * DOFS + k] = v;
        }
    __syncthreads();
    if (tid < NV) part[(int64_t)cell * NV + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

template <int DIM>
__global__ void k_restrict_nodes(CoarseGeom cg, int dofs, int64_t n_lat, const double* __restrict__ part,
                                 double* __restrict__ r0) {
    constexpr int NC = 1 << DIM;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lat * dofs) return;
    const int64_t I = t / dofs;
    const int k = (int)(t - I * dofs);
    int iI[3] = {0, 0, 0};
    {
        int64_t r = I;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            iI[d] = (int)(r % cg.np[d]);
            r /= cg.np[d];
        }
    }
    double sum = 0.0;
    for (int a = 0; a < NC; ++a) {
        int cell = 0, mul = 1;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            const int cd = iI[d] - ((a >> d) & 1);
            ok = ok && cd >= 0 && cd < cg.g[d];
            cell += mul * cd;
            mul *= cg.g[d];
        }
        if (ok) sum += part[((int64_t)cell * NC + a) * dofs + k];
    }
    r0[t] = sum;
}

// y = K x for the dense n0 x n0 matrix, one wave per row
__global__ __launch_bounds__(256) void k_dense_mv(const double* __restrict__ K, int64_t ld, int64_t n0,
                                                  const double* __restrict__ x, double* __restrict__ y) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n0) return;
    const double* __restrict__ kr = K + row * ld;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t j = lane;
    for (; j + 192 < n0; j += 256) {
        s0 = fma(kr[j], x[j], s0);
        s1 = fma(kr[j + 64], x[j + 64], s1);
        s2 = fma(kr[j + 128], x[j + 128], s2);
        s3 = fma(kr[j + 192], x[j + 192], s3);
    }
    for (; j < n0; j += 64) s0 = fma(kr[j], x[j], s0);
    double v = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) y[row] = v;
}

template <int DIM>
__global__ void k_prolong_add(CoarseGeom cg, int dofs, int64_t n_rows, const double* __restrict__ xyz,
                              const double* __restrict__ mask, const double* __restrict__ z0,
                              double* __restrict__ z) {
    constexpr int NC = 1 << DIM;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const double m = mask[row];
    if (m == 0.0) return;
    const int32_t node = (int32_t)(row / dofs);
    const int k = (int)(row - (int64_t)node * dofs);
    const Loc o = locate<DIM>(cg, xyz, node);
    double sum = 0.0;
#pragma unroll
    for (int a = 0; a < NC; ++a)
        sum = fma(corner_weight<DIM>(o, a), z0[(int64_t)lattice_node<DIM>(cg, o.i0, a) * dofs + k], sum);
    z[row] += sum * m;
}

}  // namespace

#define COARSE_DIM(KERNEL, ...)                      \
    do {                                             \
        if (dim == 3) { KERNEL(3, __VA_ARGS__); }    \
        else { KERNEL(2, __VA_ARGS__); }             \
    } while (0)

int coarse_setup(fedd_ctx* c) {
    ScopedTimer timer(c, FEDD_T_COARSE_SETUP);
    const int dim = c->dim, dofs = c->dofs;
    const int32_t n_own = (int32_t)c->n_own;
    FEDD_CHECK(dim == 2 || dim == 3, "coarse setup: dim %d", dim);
    FEDD_CHECK(dofs >= 1 && dofs <= MAX_DOFS, "coarse setup: %d dofs per node", dofs);
    FEDD_CHECK(c->n_rows == (int64_t)dofs * c->n_own, "coarse setup: the system is not node-interleaved");
    // ---- global bounding box and node count ----
    double lo[3], hi[3];
    double n_global = (double)n_own;
    FEDD_TRY(global_box(c, n_own, lo, hi, &n_global));
    // ---- lattice: g_d = max(1, floor(L_d / H + 0.5)), H = (V / cells_target)^(1/dim) ----
    double target = c->co_cells_target;
    // default: one cell per 1000 nodes, at most 12^3 cells (K0 is dense and replicated: 2197^2 doubles
    // = 38 MB to all-reduce and 21 GFLOP to invert per setup; 15^3 cells would be 134 MB and 137 GFLOP)
    if (!(target > 0)) target = std::min(1728.0, std::max(1.0, std::floor(n_global / 1000.0)));
    CoarseGeom cg;
    cg.dim = dim;
    double V = 1.0;
    for (int d = 0; d < dim; ++d) V *= (hi[d] - lo[d] > 0 ? hi[d] - lo[d] : 1.0);
    const double H = std::pow(V / target, 1.0 / dim);
    int64_t ncell = 1, nlat = 1;
    for (int d = 0; d < 3; ++d) {
        cg.g[d] = 1;
        cg.np[d] = 1;
        cg.lo[d] = 0.0;
        cg.L[d] = 1.0;
        if (d >= dim) continue;
        const double Ld = hi[d] - lo[d];
        cg.lo[d] = lo[d];
        cg.L[d] = Ld > 0 ? Ld : 1.0;
        int g = (int)std::floor(cg.L[d] / H + 0.5);
        if (g < 1 || !(Ld > 0)) g = 1;
        cg.g[d] = g;
        cg.np[d] = g + 1;
        ncell *= g;
        nlat *= g + 1;
    }
    const int64_t n0 = nlat * dofs;
    FEDD_CHECK(n0 <= COARSE_MAX_DOFS,
               "coarse setup: %lld coarse dofs, the dense coarse solver takes at most %d; lower "
               "fedd_schwarz_set_coarse (now %g cells)", (long long)n0, COARSE_MAX_DOFS, target);
    const int64_t ld = (n0 + NB - 1) / NB * NB;
    c->co_geom = cg;
    c->co_ncell = ncell;
    c->co_nlat = nlat;
    c->co_n0 = n0;
    c->co_ld = ld;
    const dim3 blk(256), gn((n_own + 255) / 256);
    // ---- owned nodes grouped by cell, in node order (stable radix split on the cell id) ----
    for (int q = 0; q < 2; ++q) {
        FEDD_TRY(c->d_co_key[q].ensure((size_t)n_own));
        FEDD_TRY(c->d_co_val[q].ensure((size_t)n_own));
    }
    FEDD_TRY(c->d_co_cell_ptr.ensure((size_t)ncell + 1));
    FEDD_TRY(c->d_itmp0.ensure((size_t)n_own));
#define K_CELL_KEY(D, ...) hipLaunchKernelGGL(k_cell_key<D>, gn, blk, 0, c->stream, __VA_ARGS__)
    COARSE_DIM(K_CELL_KEY, cg, (const double*)c->d_xyz.p, n_own, c->d_co_key[0].p, c->d_co_val[0].p);
#undef K_CELL_KEY
    int cur = 0;
    for (int bit = 0; ((int64_t)1 << bit) < ncell; ++bit) {
        hipLaunchKernelGGL(k_split_flags, gn, blk, 0, c->stream, (const int32_t*)c->d_co_key[cur].p, n_own, bit, c->d_itmp0.p);
        FEDD_TRY(exclusive_scan_i32(c, c->d_itmp0.p, c->d_itmp0.p, n_own, nullptr));
        hipLaunchKernelGGL(k_split_scatter, gn, blk, 0, c->stream, (const int32_t*)c->d_co_key[cur].p,
                           (const int32_t*)c->d_co_val[cur].p, (const int32_t*)c->d_itmp0.p, n_own, bit,
                           c->d_co_key[1 - cur].p, c->d_co_val[1 - cur].p);
        cur = 1 - cur;
    }
    c->co_sorted = cur;
    hipLaunchKernelGGL(k_cell_bounds, gn, blk, 0, c->stream, (const int32_t*)c->d_co_key[cur].p, n_own, (int32_t)ncell,
                       c->d_co_cell_ptr.p);
    const int32_t* cell_nodes = c->d_co_val[cur].p;
    // ---- Dirichlet mask over the column space (ghost dofs through the halo) ----
    FEDD_TRY(c->d_co_mask.ensure((size_t)c->n_cols));
    FEDD_TRY(c->d_flags.ensure(16));
    int32_t* d_bad = c->d_flags.p + 3;  // [0] non-neighbour coupling, [1] K0 not definite, [2] free dofs
    FEDD_HIP(hipMemsetAsync(d_bad, 0, 3 * sizeof(int32_t), c->stream));
    hipLaunchKernelGGL(k_mask, dim3((unsigned)((c->n_rows + 255) / 256)), blk, 0, c->stream, (const int32_t*)c->d_isdir.p,
                       c->n_rows, c->d_co_mask.p, d_bad + 2);
    if (c->n_cols != c->n_rows || !c->halo.peers.empty()) FEDD_TRY(halo_import(c, c->d_co_mask.p, dofs));
    // ---- K0 = Phi^T A Phi ----
    const int NC = 1 << dim, NS = dim == 3 ? 64 : 16, NW = dim == 3 ? 125 : 25;
    // chunks per cell: about 256 nodes each (4 batches), so that few, well-filled cells still
    // spread over the device
    const int nch = (int)std::min<int64_t>(16, std::max<int64_t>(1, ((int64_t)n_own / ncell + 255) / 256));
    FEDD_TRY(c->d_co_cellK.ensure((size_t)ncell * nch * dofs * dofs * NC * NS));
    FEDD_TRY(c->d_co_K.ensure((size_t)ld * ld));
    FEDD_HIP(hipMemsetAsync(c->d_co_K.p, 0, (size_t)ld * ld * sizeof(double), c->stream));
#define K_GALERKIN(D, ...) \
    hipLaunchKernelGGL(k_cell_galerkin<D>, dim3((unsigned)ncell, (unsigned)(dofs * dofs), (unsigned)nch), dim3(64), 0, c->stream, __VA_ARGS__)
    COARSE_DIM(K_GALERKIN, cg, (const int32_t*)c->d_co_cell_ptr.p, cell_nodes, (const double*)c->d_xyz.p,
               (const int32_t*)c->d_rowptr.p, (const int32_t*)c->d_colind.p, (const double*)c->d_val.p,
               (const double*)c->d_co_mask.p, dofs, c->d_co_cellK.p, d_bad);
#undef K_GALERKIN
#define K_GATHER(D, ...) hipLaunchKernelGGL(k_coarse_gather<D>, dim3((unsigned)((nlat * NW + 255) / 256)), blk, 0, c->stream, __VA_ARGS__)
    COARSE_DIM(K_GATHER, cg, dofs, nch, nlat, (const double*)c->d_co_cellK.p, c->d_co_K.p, ld);
#undef K_GATHER
    if (c->nranks > 1) {
        FEDD_CHECK(ld * ld < ((int64_t)1 << 31), "coarse setup: K0 too large for one all-reduce");
        FEDD_TRY(allreduce_sum(c, c->d_co_K.p, (int)(ld * ld)));
    }
    hipLaunchKernelGGL(k_fix_diag, dim3((unsigned)((ld + 3) / 4)), blk, 0, c->stream, c->d_co_K.p, ld, n0);
    // ---- K0 <- K0^-1 ----
    FEDD_TRY(dense_invert_batched(c, c->d_co_K.p, ld, 1, ld * ld, nullptr, (int)(ld / NB), 1, d_bad + 1));
    int32_t bad[3] = {0, 0, 0};
    FEDD_HIP(hipMemcpyAsync(bad, d_bad, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    double n_free = (double)bad[2];
    if (c->nranks > 1) {
        // every rank must take the same decision: share the flags (and add up the free dofs)
        double hb[3] = {(double)bad[0], (double)bad[1], n_free};
        FEDD_TRY(c->d_co_r0.ensure(std::max<size_t>(3, c->d_co_r0.cap)));
        FEDD_HIP(hipMemcpyAsync(c->d_co_r0.p, hb, sizeof(hb), hipMemcpyHostToDevice, c->stream));
        FEDD_TRY(allreduce_sum(c, c->d_co_r0.p, 3));
        FEDD_HIP(hipMemcpyAsync(hb, c->d_co_r0.p, sizeof(hb), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        bad[0] = hb[0] != 0.0;
        bad[1] = hb[1] != 0.0;
        n_free = hb[2];
    }
    // more coarse dofs than free fine dofs: K0 = Phi^T A Phi cannot have full rank, its regularised
    // inverse would amplify rounding errors by 1e12 and the operator would stop being linear
    FEDD_CHECK(n_free == 0.0 || n_free >= (double)n0,
               "coarse setup: %lld coarse dofs for %.0f free dofs, the lattice is too fine; lower "
               "fedd_schwarz_set_coarse (now %g cells)", (long long)n0, n_free, target);
    FEDD_CHECK(!bad[0], "coarse setup: a matrix entry couples lattice cells that are not neighbours "
                        "(the lattice is finer than the mesh); lower fedd_schwarz_set_coarse (now %g cells)", target);
    FEDD_CHECK(!bad[1], "coarse setup: K0 is not positive definite (lattice too fine for the mesh?); lower "
                        "fedd_schwarz_set_coarse (now %g cells)", target);
    FEDD_TRY(c->d_co_part.ensure((size_t)ncell * NC * dofs));
    FEDD_TRY(c->d_co_r0.ensure(std::max<size_t>((size_t)ld, c->d_co_r0.cap)));
    FEDD_TRY(c->d_co_z0.ensure((size_t)ld));
    FEDD_HIP(hipGetLastError());
    c->have_coarse = true;
    return 0;
}

int coarse_apply_add(fedd_ctx* c, const double* d_r_owned, double* d_z_owned) {
    ScopedTimer timer(c, FEDD_T_COARSE_APPLY);
    const int dim = c->dim, dofs = c->dofs;
    const CoarseGeom cg = c->co_geom;
    const int64_t ncell = c->co_ncell, nlat = c->co_nlat, n0 = c->co_n0, ld = c->co_ld;
    const int32_t* cell_nodes = c->d_co_val[c->co_sorted].p;
    const dim3 blk(256);
#define K_RESTRICT(D, F)                                                                                          \
    hipLaunchKernelGGL((k_restrict_cells<D, F>), dim3((unsigned)ncell), blk, 0, c->stream, cg,                   \
                       (const int32_t*)c->d_co_cell_ptr.p, cell_nodes, (const double*)c->d_xyz.p,                 \
                       (const double*)c->d_co_mask.p, d_r_owned, c->d_co_part.p)
    if (dim == 3) {
        if (dofs == 1) K_RESTRICT(3, 1);
        else if (dofs == 2) K_RESTRICT(3, 2);
        else K_RESTRICT(3, 3);
    } else {
        if (dofs == 1) K_RESTRICT(2, 1);
        else if (dofs == 2) K_RESTRICT(2, 2);
        else K_RESTRICT(2, 3);
    }
#undef K_RESTRICT
#define K_RNODES(D, ...) hipLaunchKernelGGL(k_restrict_nodes<D>, dim3((unsigned)((n0 + 255) / 256)), blk, 0, c->stream, __VA_ARGS__)
    COARSE_DIM(K_RNODES, cg, dofs, nlat, (const double*)c->d_co_part.p, c->d_co_r0.p);
#undef K_RNODES
    if (c->nranks > 1) {  // the collective is not part of the kernel time
        timer.stop();
        FEDD_TRY(allreduce_sum(c, c->d_co_r0.p, (int)n0));
        timer.resume();   // dense_mv and prolongation belong to the same apply (one sampling decision)
    }
    hipLaunchKernelGGL(k_dense_mv, dim3((unsigned)((n0 + 3) / 4)), blk, 0, c->stream, (const double*)c->d_co_K.p, ld, n0,
                       (const double*)c->d_co_r0.p, c->d_co_z0.p);
#define K_PROLONG(D, ...) hipLaunchKernelGGL(k_prolong_add<D>, dim3((unsigned)((c->n_rows + 255) / 256)), blk, 0, c->stream, __VA_ARGS__)
    COARSE_DIM(K_PROLONG, cg, dofs, c->n_rows, (const double*)c->d_xyz.p, (const double*)c->d_co_mask.p,
               (const double*)c->d_co_z0.p, d_z_owned);
#undef K_PROLONG
    FEDD_HIP(hipGetLastError());
    return 0;
}

}  // namespace fedd
