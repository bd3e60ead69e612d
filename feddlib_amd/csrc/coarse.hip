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
#include <climits>
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
                        // the entries of column component l, four at a time: their columns and values, then the masks and
                        // coordinates of the column nodes, requested together; the contributions are added in entry order (one
                        // entry per trip waited for two dependent loads each time: 4.9 ms for the 857 375 node rows of cfg 5's share)
                        constexpr int U = 4;
                        const int32_t pe = rowptr[row + 1];
                        for (int32_t p0 = rowptr[row]; p0 < pe; p0 += U) {
                            int32_t col[U];
                            double av[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const int32_t p = min(p0 + u, pe - 1);
                                col[u] = colind[p];
                                av[u] = val[p];
                            }
                            bool on[U];
                            double mc[U], X[U][3];
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const int32_t jn = col[u] / dofs;
                                on[u] = p0 + u < pe && col[u] - jn * dofs == l;
                                mc[u] = mask[col[u]];
#pragma unroll
                                for (int d = 0; d < DIM; ++d) X[u][d] = xyz[(int64_t)jn * DIM + d];
                            }
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const double v = av[u] * mc[u];
                                if (!on[u] || v == 0.0) continue;
                                Loc oj;
#pragma unroll
                                for (int d = 0; d < 3; ++d) {
                                    oj.i0[d] = 0;
                                    oj.f[d] = 0.0;
                                }
#pragma unroll
                                for (int d = 0; d < DIM; ++d) {     // (locate() on the coordinates already here)
                                    const double t = (X[u][d] - cg.lo[d]) / cg.L[d] * (double)cg.g[d];
                                    int i = (int)floor(t);
                                    i = min(cg.g[d] - 1, max(0, i));
                                    oj.i0[d] = i;
                                    oj.f[d] = t - (double)i;
                                }
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


// =====================================================================================================
// GDSW coarse space (coarse_kind FEDD_COARSE_GDSW): FROSch's GDSWCoarseOperator
// (feddlib/problems/tests/laplace/parametersPrec.xml:13-23, 62-122; handed over by Preconditioner_def.hpp:379-423),
// on a second, coarse decomposition: the cells of the lattice that also carries the Q1 space.
//   * every element belongs to the lattice cell of its centroid; the cells a node's elements lie in span the index
//     box [imin, imax]; the node's interface ENTITY has the coordinates e_d = imin_d + imax_d in the doubled lattice
//     (0 .. 2 g_d - 2): e_d odd = "between cells i and i + 1 in direction d".  All e_d even: interior of cell e / 2.
//     One odd e_d: face, two: edge, three: vertex (3D) -- the interface components GDSW classifies;
//   * Phi_Gamma: one function per (entity, null-space vector): the null-space vector restricted to the entity.  The null
//     space is what FROSch uses without node coordinates ("Use node lists" = false, laplace/parametersPrec.xml:5):
//     the constant per dof component (translations);
//   * Phi_I = -K_II^-1 K_IGamma Phi_Gamma, the discrete harmonic (energy-minimising) extension into the cell
//     interiors: the interiors of different cells are not coupled, and a cell sees exactly one entity of each of the
//     3^dim - 1 CLASSES (per direction: e_d even / e_d = 1 mod 4 / e_d = 3 mod 4), so one constrained solve per
//     (class, component) extends all entities of the class at once.  The solves run on the device with the
//     library's own GMRES + one-level Schwarz on the constrained operator (gmres.hip, gm_mask) to `gdsw_tol`
//     (default 1e-4): an iterative interior solver in place of FROSch's direct ExtensionSolver;
//   * K0 = Phi^T K Phi, column by colour: entities whose coordinates agree modulo 5 (RGDSW: coarse nodes, modulo 6) in every
//     direction have supports that no row couples, so one prolongation - SpMV - restriction gives one column of K0 for all of them;
//     K0 is inverted by the matrix-core sweep of dense.hip and replicated, exactly like the Q1 level.
// Storage: Phi[row][slot], slot = class * dofs + k, 3^dim - 1 classes: for the rows of cell h the entry belongs to the
// entity of that class that h sees; an interface row keeps its single 1 in the class of its own entity (home cell
// h = floor(e / 2) sees it).
// =====================================================================================================
template <int DIM> struct GdswCfg { static constexpr int NCLS = DIM == 3 ? 26 : 8; };

// The classes that carry coarse functions, numbered compactly: GDSW all 3^dim - 1, RGDSW the classes of the coarse nodes only
// (2^dim of them where every direction has >= 2 cells).  The basis is stored slot-major over these ACTIVE slots only:
// Phi_T[a][row], a = active class index * dofs + component, leading dimension ldp >= n_rows: every kernel below reads it with
// consecutive lanes on consecutive rows (the row-major [row][78] layout of round 2 made each lane walk its own 624-byte line:
// 3.5 ms per application at 2.6 M dofs), and RGDSW moves 24 instead of 78 columns.
struct GdAct {
    int n;             // active classes
    int8_t cls[26];    // active index -> class
    int8_t idx[26];    // class -> active index, -1: carries no function
};

// class code of an entity coordinate: 0 even, 1 = 1 mod 4, 2 = 3 mod 4
__device__ __forceinline__ int gd_code(int e) { return (e & 1) ? ((e & 3) == 1 ? 1 : 2) : 0; }

// class index 0 .. 3^DIM - 2 of entity e (-1: cell interior)
template <int DIM>
__device__ __forceinline__ int gd_class(const int e[3]) {
    int s = 0, mul = 1;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        s += mul * gd_code(e[d]);
        mul *= 3;
    }
    return s - 1;
}

// entity of class `cls` seen from cell h; false if it lies outside the lattice
template <int DIM>
__device__ __forceinline__ bool gd_entity_of(const CoarseGeom& cg, const int h[3], int cls, int e[3]) {
    int s = cls + 1;
    bool ok = true;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        const int code = s % 3;
        s /= 3;
        if (code == 0) e[d] = 2 * h[d];
        else e[d] = ((h[d] & 1) == 0) == (code == 1) ? 2 * h[d] + 1 : 2 * h[d] - 1;
        ok = ok && e[d] >= 0 && e[d] <= 2 * cg.g[d] - 2;
    }
    return ok;
}

template <int DIM>
__device__ __forceinline__ int32_t gd_entity_id(const CoarseGeom& cg, const int e[3]) {
    int32_t id = 0, mul = 1;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        id += mul * e[d];
        mul *= 2 * cg.g[d] - 1;
    }
    return id;
}

template <int DIM>
__device__ __forceinline__ void gd_entity_coords(const CoarseGeom& cg, int32_t id, int e[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) e[d] = 0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        const int m = 2 * cg.g[d] - 1;
        e[d] = id % m;
        id /= m;
    }
}

// Coarse dof numbering.  GDSW: every entity of the doubled lattice is a coarse "node" (id = entity id; cell interiors
// and empty entities get a unit diagonal in K0).  RGDSW (cg.reduced): only the COARSE NODES -- the entities with an odd
// coordinate in every direction that has more than one cell (the vertices of the decomposition; for slab / pencil
// decompositions the faces / edges that have no lower-dimensional neighbour) -- numbered compactly:
// id = sum_d ((e_d - 1) / 2) * stride over the directions with g_d >= 2.
template <int DIM>
__device__ __forceinline__ bool gd_is_coarse_node(const CoarseGeom& cg, const int e[3]) {
    bool ok = true;
#pragma unroll
    for (int d = 0; d < DIM; ++d) ok = ok && (cg.g[d] >= 2 ? (e[d] & 1) == 1 : e[d] == 0);
    return ok;
}

template <int DIM>
__device__ __forceinline__ int64_t gd_coarse_count(const CoarseGeom& cg) {
    int64_t n = 1;
#pragma unroll
    for (int d = 0; d < DIM; ++d) n *= cg.reduced ? (cg.g[d] >= 2 ? cg.g[d] - 1 : 1) : 2 * cg.g[d] - 1;
    return n;
}

// coarse id of entity e, -1 if it carries no coarse dof
template <int DIM>
__device__ __forceinline__ int32_t gd_coarse_id(const CoarseGeom& cg, const int e[3]) {
    if (!cg.reduced) return gd_entity_id<DIM>(cg, e);
    if (!gd_is_coarse_node<DIM>(cg, e)) return -1;
    int32_t id = 0, mul = 1;
#pragma unroll
    for (int d = 0; d < DIM; ++d)
        if (cg.g[d] >= 2) {
            id += mul * ((e[d] - 1) >> 1);
            mul *= cg.g[d] - 1;
        }
    return id;
}

// entity coordinates of coarse id
template <int DIM>
__device__ __forceinline__ void gd_coarse_coords(const CoarseGeom& cg, int32_t id, int e[3]) {
    if (!cg.reduced) {
        gd_entity_coords<DIM>(cg, id, e);
        return;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) e[d] = 0;
#pragma unroll
    for (int d = 0; d < DIM; ++d)
        if (cg.g[d] >= 2) {
            const int m = cg.g[d] - 1;
            e[d] = 2 * (id % m) + 1;
            id /= m;
        }
}

// entity and home cell of every owned node from the cells of its incident elements
template <int DIM>
__global__ void k_gd_node_entity(CoarseGeom cg, const int32_t* __restrict__ conn, int nen, const double* __restrict__ xyz,
                                 const int32_t* __restrict__ n2e_ptr, const int32_t* __restrict__ n2e, int32_t n_own,
                                 int32_t* __restrict__ ent, int32_t* __restrict__ key, int32_t* __restrict__ val) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_own) return;
    int imin[3] = {INT_MAX, INT_MAX, INT_MAX}, imax[3] = {-1, -1, -1};
    for (int32_t p = n2e_ptr[i]; p < n2e_ptr[i + 1]; ++p) {
        const int32_t e = n2e[p] / nen;
        double cen[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) cen[d] = 0.0;
        for (int v = 0; v <= DIM; ++v) {
            const int32_t nd = conn[(int64_t)e * nen + v];
#pragma unroll
            for (int d = 0; d < DIM; ++d) cen[d] += xyz[(int64_t)nd * DIM + d];
        }
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            const double t = (cen[d] / (DIM + 1) - cg.lo[d]) / cg.L[d] * (double)cg.g[d];
            int ic = (int)floor(t);
            ic = min(cg.g[d] - 1, max(0, ic));
            imin[d] = min(imin[d], ic);
            imax[d] = max(imax[d], ic);
        }
    }
    int e3[3] = {0, 0, 0}, h[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        if (imax[d] < 0) imin[d] = imax[d] = 0;   // a node without elements: interior of cell 0
        e3[d] = imin[d] + imax[d];
        h[d] = e3[d] >> 1;
    }
    ent[i] = gd_entity_id<DIM>(cg, e3);
    key[i] = cell_of<DIM>(cg, h);
    val[i] = i;
}

// The coarse functions an interface node of entity e (class cls >= 0) takes part in: GDSW the functions of e itself; RGDSW,
// option 1 (Dohrmann, Widlund 2017): those of every coarse node of C(e), the coarse nodes adjacent to e -- e_d odd stays, an
// even e_d (direction with >= 2 cells) moves to e_d - 1 or e_d + 1 where that lies inside the lattice; all of them are
// corners of the home cell -- each with the weight 1 / |C(e)|.  Returns their number; ai = active class index, E = coarse id,
// ev = entity coordinates of the function's own entity (the centre of its rotations).
template <int DIM>
__device__ __forceinline__ int gd_targets(const CoarseGeom& cg, const GdAct& act, const int e[3], int cls, int ai[8], int32_t E[8],
                                          int ev[8][3]) {
    constexpr int NCLS = GdswCfg<DIM>::NCLS;
    if (!cg.reduced) {
        if (act.idx[cls] < 0) return 0;
        ai[0] = act.idx[cls];
        E[0] = gd_entity_id<DIM>(cg, e);
#pragma unroll
        for (int d = 0; d < 3; ++d) ev[0][d] = d < DIM ? e[d] : 0;
        return 1;
    }
    int h[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < DIM; ++d) h[d] = e[d] >> 1;
    int count = 0;
    for (int c2 = 0; c2 < NCLS; ++c2) {
        int v[3] = {0, 0, 0};
        if (act.idx[c2] < 0 || !gd_entity_of<DIM>(cg, h, c2, v) || !gd_is_coarse_node<DIM>(cg, v)) continue;
        bool adj = true;
#pragma unroll
        for (int d = 0; d < DIM; ++d) adj = adj && ((e[d] & 1) ? v[d] == e[d] : (cg.g[d] >= 2 ? (v[d] == e[d] - 1 || v[d] == e[d] + 1) : v[d] == e[d]));
        if (!adj || count >= 8) continue;
        ai[count] = act.idx[c2];
        E[count] = gd_coarse_id<DIM>(cg, v);
#pragma unroll
        for (int d = 0; d < 3; ++d) ev[count][d] = v[d];
        ++count;
    }
    return count;
}

// Null space of the operator restricted to an entity, function k of nns, component a at offset dd from the entity's centre:
// k < dofs the translations (constant 1 in component k: all FROSch uses without node coordinates), k >= dofs the linearised
// rotations FROSch adds with "Use node lists" and "Rotations" = true (steadyLinElas/parametersPrec.xml:6, 100): 2D (-y, x);
// 3D about z (-y, x, 0), about x (0, -z, y), about y (z, 0, -x).  The centre is that of the entity in the lattice,
// lo_d + (e_d + 1) H_d / 2: translations + rotations span the same space whatever the centre, this one keeps the rotations of
// single-node vertices and the axial rotation of straight edges at zero, where the selection below drops them.
template <int DIM>
__device__ __forceinline__ double gd_null(int dofs, int k, int a, const double dd[3]) {
    if (k < dofs) return k == a ? 1.0 : 0.0;
    const int j = k - dofs;
    if (DIM == 2) return a == 0 ? -dd[1] : dd[0];
    if (j == 0) return a == 0 ? -dd[1] : (a == 1 ? dd[0] : 0.0);
    if (j == 1) return a == 0 ? 0.0 : (a == 1 ? -dd[2] : dd[1]);
    return a == 0 ? dd[2] : (a == 1 ? 0.0 : -dd[0]);
}

template <int DIM>
__device__ __forceinline__ void gd_offset(const CoarseGeom& cg, const double* __restrict__ xyz, int32_t node, const int ev[3], double dd[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) dd[d] = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) dd[d] = xyz[(int64_t)node * DIM + d] - (cg.lo[d] + (double)(ev[d] + 1) * (cg.L[d] / (double)cg.g[d]) * 0.5);
}

constexpr int GD_NNS_MAX = 6;                                   // 3 translations + 3 rotations
constexpr int GD_NG = GD_NNS_MAX * (GD_NNS_MAX + 1) / 2;        // packed lower triangle of an entity's Gram matrix

// Rotations: which of an entity's nns functions are linearly independent on its free interface dofs (FROSch drops the
// dependent ones: no rotations on vertices and one-node edges, two on straight edges).  Step 1: the Gram matrix of the functions
// per coarse id, G[E][k (k + 1) / 2 + l] = sum over interface nodes and free components of f_k f_l (atomics: the sums only
// feed the thresholds of step 2).
template <int DIM>
__global__ void k_gd_gram(CoarseGeom cg, GdAct act, const int32_t* __restrict__ ent, int dofs, int nns, int32_t n_own,
                          const double* __restrict__ xyz, const double* __restrict__ mask, double* __restrict__ G) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_own) return;
    int e[3];
    gd_entity_coords<DIM>(cg, ent[i], e);
    const int cls = gd_class<DIM>(e);
    if (cls < 0) return;
    int ai[8], ev[8][3];
    int32_t E[8];
    const int nt = gd_targets<DIM>(cg, act, e, cls, ai, E, ev);
    for (int t = 0; t < nt; ++t) {
        double dd[3];
        gd_offset<DIM>(cg, xyz, i, ev[t], dd);
        const double w = 1.0 / (double)nt;
        for (int k = 0; k < nns; ++k)
            for (int l = 0; l <= k; ++l) {
                double sum = 0.0;
                for (int a = 0; a < dofs; ++a) {
                    const double m = mask[(int64_t)i * dofs + a];
                    sum += (w * gd_null<DIM>(dofs, k, a, dd) * m) * (w * gd_null<DIM>(dofs, l, a, dd) * m);
                }
                if (sum != 0.0) atomicAdd(&G[(int64_t)E[t] * GD_NG + k * (k + 1) / 2 + l], sum);
            }
    }
}

// Step 2: a Cholesky sweep over the functions in their order (translations first): function k stays if what is left of it after
// the kept ones before it exceeds 1e-8 x (its scale) x (the largest translation's sum): scale 1 for a translation, Hmax^2 for a
// rotation.  keep[E] = bit mask of the functions kept.
__global__ void k_gd_select(const double* __restrict__ G, int64_t n_coarse, int dofs, int nns, double hmax2, int32_t* __restrict__ keep) {
    const int64_t E = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (E >= n_coarse) return;
    double g[GD_NNS_MAX][GD_NNS_MAX], L[GD_NNS_MAX][GD_NNS_MAX];
    for (int k = 0; k < nns; ++k)
        for (int l = 0; l <= k; ++l) {
            g[k][l] = G[E * GD_NG + k * (k + 1) / 2 + l];
            L[k][l] = 0.0;
        }
    double tmax = 0.0;
    for (int t = 0; t < dofs; ++t) tmax = fmax(tmax, g[t][t]);
    int32_t bits = 0;
    for (int k = 0; k < nns; ++k) {
        double r = g[k][k];
        for (int j = 0; j < k; ++j) {
            if (!((bits >> j) & 1)) continue;
            double v = g[k][j];
            for (int q = 0; q < j; ++q)
                if ((bits >> q) & 1) v -= L[k][q] * L[j][q];
            L[k][j] = v / L[j][j];
            r -= L[k][j] * L[k][j];
        }
        if (tmax > 0.0 && r > 1e-8 * (k < dofs ? 1.0 : hmax2) * tmax) {
            bits |= 1 << k;
            L[k][k] = sqrt(r);
        }
    }
    keep[E] = bits;
}

// Phi <- Phi_Gamma (interface rows: the null-space functions of their own entity -- RGDSW: of the adjacent coarse nodes, weighted --
// in the class of that entity; Dirichlet rows 0), interior rows 0; imask = 1 on free interior dofs (the unknowns of the
// extension solves), 0 elsewhere.  nns functions per entity (dofs translations, then the rotations if any); keep: the bit
// masks of k_gd_select, nullptr = all kept.
template <int DIM>
__global__ void k_gd_phi_init(CoarseGeom cg, GdAct act, const int32_t* __restrict__ ent, int dofs, int nns, int64_t n_rows, int64_t ldp,
                              const double* __restrict__ xyz, const int32_t* __restrict__ keep,
                              const double* __restrict__ mask, double* __restrict__ phiT, double* __restrict__ imask) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int32_t node = (int32_t)(r / dofs);
    const int a = (int)(r - (int64_t)node * dofs);
    int e[3];
    gd_entity_coords<DIM>(cg, ent[node], e);
    const int cls = gd_class<DIM>(e);
    const int nsa = act.n * nns;
    for (int s = 0; s < nsa; ++s) phiT[(int64_t)s * ldp + r] = 0.0;
    if (cls >= 0) {
        int ai[8], ev[8][3];
        int32_t E[8];
        const int nt = gd_targets<DIM>(cg, act, e, cls, ai, E, ev);
        for (int t = 0; t < nt; ++t) {
            double dd[3] = {0.0, 0.0, 0.0};
            if (nns > dofs) gd_offset<DIM>(cg, xyz, node, ev[t], dd);
            const int32_t bits = keep ? keep[E[t]] : -1;
            for (int k = 0; k < nns; ++k)
                if ((bits >> k) & 1) phiT[(int64_t)(ai[t] * nns + k) * ldp + r] = gd_null<DIM>(dofs, k, a, dd) * mask[r] / (double)nt;
        }
    }
    imask[r] = cls < 0 ? mask[r] : 0.0;
}

// v = one column of Phi on the interface rows, 0 on the interior rows
__global__ void k_gd_gamma_col(const double* __restrict__ phi_col, const double* __restrict__ imask, int64_t n_rows,
                               const double* __restrict__ mask, double* __restrict__ v) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    v[r] = (imask[r] == 0.0 && mask[r] != 0.0) ? phi_col[r] : 0.0;
}

// b = -imask o w
__global__ void k_gd_rhs(const double* __restrict__ imask, const double* __restrict__ w, int64_t n_rows, double* __restrict__ b) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows) b[r] = imask[r] != 0.0 ? -w[r] : 0.0;
}

// interior rows: Phi column = x
__global__ void k_gd_store(const double* __restrict__ imask, const double* __restrict__ x, int64_t n_rows, double* __restrict__ phi_col) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows && imask[r] != 0.0) phi_col[r] = x[r];
}

// ... the same three steps for MULTI_NR columns at a time on stacked vectors X[r * MULTI_NR + j] (multi.hip): column j is slot
// slot0 + j, columns j >= nb are zero
__global__ void k_gd_gamma_cols(const double* __restrict__ phiT, int64_t ldp, int slot0, int nb, const double* __restrict__ imask,
                                int64_t n_rows, const double* __restrict__ mask, double* __restrict__ V) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t / MULTI_NR;
    const int j = (int)(t % MULTI_NR);
    if (r >= n_rows) return;
    V[t] = (j < nb && imask[r] == 0.0 && mask[r] != 0.0) ? phiT[(int64_t)(slot0 + j) * ldp + r] : 0.0;
}

__global__ void k_gd_rhs_cols(const double* __restrict__ imask, const double* __restrict__ W, int64_t n_rows, double* __restrict__ B) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_rows * MULTI_NR) B[t] = imask[t / MULTI_NR] != 0.0 ? -W[t] : 0.0;
}

// Sixteen stacked right-hand sides are solved as ONE system, so the solver's tolerance applies to the Frobenius norm of the
// stacked residual: a column with a small right-hand side could stop at a much larger relative residual of its own (ADVICE
// r03).  The columns are therefore scaled to unit norm before the solve and back after it: every column then ends within
// sqrt(16) x the tolerance relative to ITS right-hand side, whatever shares its batch.
// k_cols_sumsq: per-column sums of squares of X[row * 16 + j], partial per workgroup; k_cols_scale_setup: norms -> scale
// factors (1 / norm, 0 for a zero column) and the norms themselves; k_cols_scale: X[row * 16 + j] *= f[j].
constexpr int CS_ROWS = 4096;       // rows per workgroup of k_cols_sumsq
__global__ __launch_bounds__(256) void k_cols_sumsq(const double* __restrict__ X, int64_t n_rows, double* __restrict__ part) {
    __shared__ double sh[16][MULTI_NR + 1];
    const int j = threadIdx.x & 15, a = threadIdx.x >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS, r1 = r0 + CS_ROWS < n_rows ? r0 + CS_ROWS : n_rows;
    double s = 0.0;
    for (int64_t r = r0 + a; r < r1; r += 16) {
        const double v = X[r * MULTI_NR + j];
        s += v * v;
    }
    sh[a][j] = s;
    __syncthreads();
    if (threadIdx.x < MULTI_NR) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += sh[q][threadIdx.x];
        part[(int64_t)blockIdx.x * MULTI_NR + threadIdx.x] = t;
    }
}
__global__ void k_cols_sum_parts(const double* __restrict__ part, int nblk, double* __restrict__ out) {   // 16 threads, fixed order
    const int j = threadIdx.x;
    if (j >= MULTI_NR) return;
    double t = 0.0;
    for (int b = 0; b < nblk; ++b) t += part[(int64_t)b * MULTI_NR + j];
    out[j] = t;
}
__global__ void k_cols_scale_setup(double* __restrict__ f) {     // in: sums of squares [16]; out: f[0..15] = 1 / norm, f[16..31] = norm
    const int j = threadIdx.x;
    if (j >= MULTI_NR) return;
    const double nrm = sqrt(f[j]);
    f[j] = nrm > 0.0 ? 1.0 / nrm : 0.0;
    f[MULTI_NR + j] = nrm;
}
__global__ void k_cols_scale(double* __restrict__ X, int64_t n, const double* __restrict__ f) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) X[t] *= f[t & (MULTI_NR - 1)];
}

// (tiles of 16 rows x 16 columns through LDS: the stacked vector is read and the Phi columns are written in whole lines)
__global__ __launch_bounds__(256) void k_gd_store_cols(const double* __restrict__ imask, const double* __restrict__ X, int64_t n_rows,
                                                       double* __restrict__ phiT, int64_t ldp, int slot0, int nb) {
    __shared__ double tile[16][MULTI_NR + 1];
    const int64_t r0 = (int64_t)blockIdx.x * 16;
    const int a = threadIdx.x >> 4, b = threadIdx.x & 15;
    if (r0 + a < n_rows) tile[a][b] = X[(r0 + a) * MULTI_NR + b];
    __syncthreads();
    const int64_t r = r0 + b;       // column a, row r0 + b
    if (r < n_rows && a < nb && imask[r] != 0.0) phiT[(int64_t)(slot0 + a) * ldp + r] = tile[b][a];
}

// restriction, step 1: per home cell the sums over its rows of Phi[r][slot] * rv[r], slot chunk `blockIdx.y` of 16,
// fixed-order workgroup reduction (same pattern as k_restrict_cells)
__global__ __launch_bounds__(256) void k_gd_restrict_cells(const int32_t* __restrict__ cell_ptr,
                                                           const int32_t* __restrict__ cell_nodes, int dofs, int nsd, int64_t ldp,
                                                           const double* __restrict__ phiT, const double* __restrict__ rv,
                                                           double* __restrict__ part) {
    constexpr int CH = 8;
    __shared__ double red[4][CH];
    const int cell = blockIdx.x, s0 = blockIdx.y * CH, tid = threadIdx.x;
    double acc[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) acc[q] = 0.0;
    const int32_t b = cell_ptr[cell], e = cell_ptr[cell + 1];
    // four rows per lane and trip: their node ids, then their entries of rv and of the CH columns of Phi, requested together and
    // added in row order (one row per trip put two dependent memory latencies in every trip: 155 us per launch with RGDSW's 24 columns
    // at cfg 5's share, where the 0.49 GB of Phi take 0.1 ms)
    constexpr int U = 4;
    const int64_t ib = (int64_t)b * dofs, ie = (int64_t)e * dofs;
    for (int64_t item0 = ib + tid; item0 < ie; item0 += 256 * U) {
        int64_t r[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t item = item0 + 256 * u;
            on[u] = item < ie;
            const int64_t it = on[u] ? item : ib;
            r[u] = (int64_t)cell_nodes[it / dofs] * dofs + it % dofs;
        }
        double x[U], ph[U][CH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = rv[r[u]];
#pragma unroll
            for (int q = 0; q < CH; ++q) ph[u][q] = phiT[(int64_t)min(s0 + q, nsd - 1) * ldp + r[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < CH; ++q)
                if (on[u] && s0 + q < nsd) acc[q] = fma(ph[u][q], x[u], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        double v = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) red[tid >> 6][q] = v;
    }
    __syncthreads();
    if (tid < CH && s0 + tid < nsd) part[(int64_t)cell * nsd + s0 + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// restriction, step 2: r0[(E, k)] = sum over the cells that see E (fixed order) of their class(E) partial sum
template <int DIM>
__global__ void k_gd_restrict_ent(CoarseGeom cg, GdAct act, int nns, int64_t n_ent, const double* __restrict__ part,
                                  double* __restrict__ r0) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_ent * nns) return;
    const int32_t E = (int32_t)(t / nns);
    const int k = (int)(t - (int64_t)E * nns);
    int e[3];
    gd_coarse_coords<DIM>(cg, E, e);
    const int cls = gd_class<DIM>(e);
    double sum = 0.0;
    if (cls >= 0 && act.idx[cls] >= 0) {
        // cells h with |2 h_d - e_d| <= 1: e_d even -> h_d = e_d / 2; odd -> (e_d - 1) / 2 and (e_d + 1) / 2
        for (int a = 0; a < (1 << DIM); ++a) {
            int h[3] = {0, 0, 0};
            bool ok = true;
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                const int bit = (a >> d) & 1;
                if (e[d] & 1) h[d] = (e[d] - 1) / 2 + bit;
                else {
                    h[d] = e[d] / 2;
                    ok = ok && bit == 0;
                }
                ok = ok && h[d] >= 0 && h[d] < cg.g[d];
            }
            if (ok) sum += part[(int64_t)cell_of<DIM>(cg, h) * (act.n * nns) + act.idx[cls] * nns + k];
        }
    }
    r0[t] = sum;
}

// prolongation: z[r] (+)= mask[r] * sum_slots Phi[slot][r] * z0[entity(home(r), class)][k]
template <int DIM, bool ADD>
__global__ void k_gd_prolong(CoarseGeom cg, GdAct act, const int32_t* __restrict__ ent, int dofs, int nns, int64_t n_rows, int64_t ldp,
                             const double* __restrict__ phiT, const double* __restrict__ mask,
                             const double* __restrict__ z0, double* __restrict__ z) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int32_t node = (int32_t)(r / dofs);
    int e[3], h[3] = {0, 0, 0};
    gd_entity_coords<DIM>(cg, ent[node], e);
#pragma unroll
    for (int d = 0; d < DIM; ++d) h[d] = e[d] >> 1;
    double sum = 0.0;
    for (int ai = 0; ai < act.n; ++ai) {
        int ee[3] = {0, 0, 0};
        if (!gd_entity_of<DIM>(cg, h, act.cls[ai], ee)) continue;
        const int64_t E = gd_coarse_id<DIM>(cg, ee);
        if (E < 0) continue;
        for (int k = 0; k < nns; ++k) sum = fma(phiT[(int64_t)(ai * nns + k) * ldp + r], z0[E * nns + k], sum);
    }
    if (ADD) z[r] += sum * mask[r];
    else z[r] = sum * mask[r];
}

struct GdCol { int c[3]; };

// Colouring of the coarse dofs for the column-wise Galerkin product: entities whose coordinates agree modulo the period
// in every direction have supports that no matrix row couples, and every entity couples with at most one of them.
// GDSW: entities couple within two steps of the doubled lattice -> period 5.  RGDSW: only the coarse nodes (odd
// coordinates) carry functions and couple with their lattice neighbours (two steps away), so same-coloured ones must be
// three coarse nodes apart: period 6, and only the odd residues occur: 3^dim colours instead of 5^dim.
__host__ __device__ __forceinline__ int gd_period(const CoarseGeom& cg) { return cg.reduced ? 6 : 5; }

// v = Phi z0 for z0 = the sum of the unit vectors (E, k) over the interface entities E of one colour (e_d = col_d mod 5):
// a cell sees three consecutive entity coordinates per direction, so at most one entity of the colour, i.e. a row reads at
// most ONE entry of Phi (the generic prolongation read all of them, 375 times per setup)
template <int DIM>
__global__ void k_gd_prolong_colour(CoarseGeom cg, GdAct act, const int32_t* __restrict__ ent, int dofs, int nns, int64_t n_rows, int64_t ldp,
                                    const double* __restrict__ phiT, const double* __restrict__ mask, int k, GdCol col,
                                    double* __restrict__ v) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int32_t node = (int32_t)(r / dofs);
    int e[3], h[3] = {0, 0, 0};
    gd_entity_coords<DIM>(cg, ent[node], e);
#pragma unroll
    for (int d = 0; d < DIM; ++d) h[d] = e[d] >> 1;
    double val = 0.0;
    for (int ai = 0; ai < act.n; ++ai) {
        int ee[3] = {0, 0, 0};
        if (!gd_entity_of<DIM>(cg, h, act.cls[ai], ee)) continue;
        bool on = gd_coarse_id<DIM>(cg, ee) >= 0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) on = on && (ee[d] % gd_period(cg)) == col.c[d];
        if (on) val = phiT[(int64_t)(ai * nns + k) * ldp + r];
    }
    v[r] = val * mask[r];
}

// K0[(E, a)][(E', k)] = r0[(E, a)] with E' the entity of the colour within two lattice steps of E
template <int DIM>
__global__ void k_gd_scatter_col(CoarseGeom cg, int nns, int64_t n_ent, int k, GdCol col, const double* __restrict__ r0,
                                 double* __restrict__ K, int64_t ld) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_ent * nns) return;
    const int32_t E = (int32_t)(t / nns);
    int e[3], ep[3] = {0, 0, 0};
    gd_coarse_coords<DIM>(cg, E, e);
    if (gd_class<DIM>(e) < 0) return;
    bool ok = true;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        const int P = gd_period(cg);
        int dlt = (col.c[d] - e[d] % P + P) % P;   // 0 .. P - 1
        if (dlt > P / 2) dlt -= P;                 // GDSW -2 .. 2; RGDSW (odd coordinates, odd colours) -2, 0, 2
        ep[d] = e[d] + dlt;
        ok = ok && ep[d] >= 0 && ep[d] <= 2 * cg.g[d] - 2;
    }
    if (!ok || gd_class<DIM>(ep) < 0) return;
    const int64_t Ep = gd_coarse_id<DIM>(cg, ep);
    if (Ep < 0) return;
    K[t * ld + Ep * nns + k] = r0[t];
}

// ---- the Galerkin product MULTI_NR columns at a time (stacked vectors X[r * MULTI_NR + j], multi.hip) ----
struct GdCols {
    int nb;
    int8_t c[MULTI_NR][3];   // colour of column j
    int8_t k[MULTI_NR];      // component of column j
};

// V[r][j] = the prolongation of the unit vectors (E, k_j) over the entities E of colour j (k_gd_prolong_colour for each column).
// A thread per row: the entities its home cell sees are worked out once, each column then only compares colours (a thread per
// (row, column) repeated the entity arithmetic sixteen times: 1.97 ms per launch with the 26 classes of GDSW).
template <int DIM>
__global__ void k_gd_prolong_colours(CoarseGeom cg, GdAct act, const int32_t* __restrict__ ent, int dofs, int nns, int64_t n_rows, int64_t ldp,
                                     const double* __restrict__ phiT, const double* __restrict__ mask, GdCols cols,
                                     double* __restrict__ V) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int32_t node = (int32_t)(r / dofs);
    int e[3], h[3] = {0, 0, 0};
    gd_entity_coords<DIM>(cg, ent[node], e);
#pragma unroll
    for (int d = 0; d < DIM; ++d) h[d] = e[d] >> 1;
    int sel[MULTI_NR];      // active class whose entity has column j's colour (a cell sees at most one), -1: none
#pragma unroll
    for (int j = 0; j < MULTI_NR; ++j) sel[j] = -1;
    const int P = gd_period(cg);
    for (int ai = 0; ai < act.n; ++ai) {
        int ee[3] = {0, 0, 0};
        if (!gd_entity_of<DIM>(cg, h, act.cls[ai], ee)) continue;
        if (gd_coarse_id<DIM>(cg, ee) < 0) continue;
        int res[3] = {0, 0, 0};
#pragma unroll
        for (int d = 0; d < DIM; ++d) res[d] = ee[d] % P;
#pragma unroll
        for (int j = 0; j < MULTI_NR; ++j) {
            bool on = true;
#pragma unroll
            for (int d = 0; d < DIM; ++d) on = on && res[d] == cols.c[j][d];
            if (on) sel[j] = ai;
        }
    }
    const double m = mask[r];
#pragma unroll
    for (int j = 0; j < MULTI_NR; ++j) {
        double val = 0.0;
        if (j < cols.nb && sel[j] >= 0) val = phiT[(int64_t)(sel[j] * nns + cols.k[j]) * ldp + r] * m;
        V[r * MULTI_NR + j] = val;
    }
}

// restriction of MULTI_NR columns, step 1: per (home cell, chunk of its rows, tile of 16 slots) the 16 x 16 block
//   part[slot][j] = sum over the chunk's rows of Phi[slot][r] W[r][j]
// on the f64 matrix cores (M = slots, N = columns, K = rows): Phi is read once for sixteen columns.  The four waves take the
// chunk's row quadruples in turn; their partial tiles are added in wave order (fixed order: reproducible).
__global__ __launch_bounds__(256) void k_gd_restrict_cells_cols(const int32_t* __restrict__ cell_ptr, const int32_t* __restrict__ cell_nodes,
                                                                int dofs, int nsd, int nch, int64_t ldp, const double* __restrict__ phiT,
                                                                const double* __restrict__ W, double* __restrict__ part) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    __shared__ double red[4][4][64];
    const int cell = blockIdx.x, st = blockIdx.y, ch = blockIdx.z;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lj = lane & 15, lk = lane >> 4;
    const int64_t items = (int64_t)(cell_ptr[cell + 1] - cell_ptr[cell]) * dofs, i0 = (int64_t)cell_ptr[cell] * dofs;
    const int64_t per = ((items + nch - 1) / nch + 3) & ~(int64_t)3;
    const int64_t b = min(items, per * ch), e = min(items, per * (ch + 1));
    const int slot = 16 * st + lj;
    const double* __restrict__ prow = phiT + (int64_t)min(slot, nsd - 1) * ldp;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int64_t q = b + 4 * w; q < e; q += 16) {
        const int64_t item = q + lk;
        const bool on = item < e;
        const int64_t it = i0 + (on ? item : b);
        const int32_t node = cell_nodes[it / dofs];
        const int64_t r = (int64_t)node * dofs + it % dofs;
        double a = prow[r], x = W[r * MULTI_NR + lj];
        a = (on && slot < nsd) ? a : 0.0;
        x = on ? x : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[w][q][lane] = acc[q];
    __syncthreads();
    // register q at lane (lk, lj): slot 16 st + lk + 4 q, column lj; wave w adds register w
    const int so = 16 * st + lk + 4 * w;
    if (so < nsd)
        part[(((int64_t)cell * nch + ch) * nsd + so) * MULTI_NR + lj] = ((red[0][w][lane] + red[1][w][lane]) + red[2][w][lane]) + red[3][w][lane];
}

// step 2: r0[(E, a)][j] = sum over the cells that see E and their chunks (fixed order) of the class(E) partial sums
template <int DIM>
__global__ void k_gd_restrict_ent_cols(CoarseGeom cg, GdAct act, int nns, int64_t n_ent, int nch, const double* __restrict__ part,
                                       double* __restrict__ r0) {
    const int64_t tt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tt >= n_ent * nns * MULTI_NR) return;
    const int j = (int)(tt % MULTI_NR);
    const int64_t t = tt / MULTI_NR;
    const int32_t E = (int32_t)(t / nns);
    const int k = (int)(t - (int64_t)E * nns);
    int e[3];
    gd_coarse_coords<DIM>(cg, E, e);
    const int cls = gd_class<DIM>(e);
    double sum = 0.0;
    if (cls >= 0 && act.idx[cls] >= 0) {
        const int nsd = act.n * nns;
        for (int a = 0; a < (1 << DIM); ++a) {
            int h[3] = {0, 0, 0};
            bool ok = true;
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                const int bit = (a >> d) & 1;
                if (e[d] & 1) h[d] = (e[d] - 1) / 2 + bit;
                else {
                    h[d] = e[d] / 2;
                    ok = ok && bit == 0;
                }
                ok = ok && h[d] >= 0 && h[d] < cg.g[d];
            }
            if (ok) {
                const int64_t cell = cell_of<DIM>(cg, h);
                for (int ch = 0; ch < nch; ++ch) sum += part[((cell * nch + ch) * nsd + act.idx[cls] * nns + k) * MULTI_NR + j];
            }
        }
    }
    r0[tt] = sum;
}

// K0[(E, a)][(E', k_j)] = r0[(E, a)][j] with E' the entity of column j's colour within two lattice steps of E
template <int DIM>
__global__ void k_gd_scatter_cols(CoarseGeom cg, int nns, int64_t n_ent, GdCols cols, const double* __restrict__ r0,
                                  double* __restrict__ K, int64_t ld) {
    const int64_t tt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tt >= n_ent * nns * MULTI_NR) return;
    const int j = (int)(tt % MULTI_NR);
    if (j >= cols.nb) return;
    const int64_t t = tt / MULTI_NR;
    const int32_t E = (int32_t)(t / nns);
    int e[3], ep[3] = {0, 0, 0};
    gd_coarse_coords<DIM>(cg, E, e);
    if (gd_class<DIM>(e) < 0) return;
    bool ok = true;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        const int P = gd_period(cg);
        int dlt = (cols.c[j][d] - e[d] % P + P) % P;
        if (dlt > P / 2) dlt -= P;
        ep[d] = e[d] + dlt;
        ok = ok && ep[d] >= 0 && ep[d] <= 2 * cg.g[d] - 2;
    }
    if (!ok || gd_class<DIM>(ep) < 0) return;
    const int64_t Ep = gd_coarse_id<DIM>(cg, ep);
    if (Ep < 0) return;
    K[t * ld + Ep * nns + cols.k[j]] = r0[tt];
}

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


// owned nodes grouped by cell, in node order within a cell: a stable radix sort of (cell, node) over the bits a cell id takes
// (the in-tree radix sort of scan.hip, 8 bits per pass; rounds 2-3: rocPRIM through hipCUB; round 1: a bit-by-bit split, three
// launches per bit, 2.5 ms for 9.9 M nodes and 1728 cells)
static int sort_nodes_by_cell(fedd_ctx* c, int32_t n_own, int64_t ncell, int* cur_out) {
    int bits = 0;
    while (((int64_t)1 << bits) < ncell) ++bits;
    *cur_out = 0;
    if (bits == 0 || n_own == 0) return 0;
    int32_t* keys[2] = {c->d_co_key[0].p, c->d_co_key[1].p};
    int32_t* vals[2] = {c->d_co_val[0].p, c->d_co_val[1].p};
    return radix_sort_pairs_i32(c, keys, vals, n_own, bits, cur_out);
}

// ---- GDSW: setup and application (kernels and definitions: "GDSW coarse space" above) ----
static GdAct gdsw_active(const CoarseGeom& cg) {
    GdAct act;
    const int dim = cg.dim, ncls = dim == 3 ? 26 : 8;
    act.n = 0;
    for (int cls = 0; cls < 26; ++cls) act.idx[cls] = -1;
    for (int cls = 0; cls < ncls; ++cls) {
        bool on = true;
        if (cg.reduced) {   // only the classes of coarse nodes carry functions: code 1 or 2 wherever there is more than one cell
            int code = cls + 1;
            for (int d = 0; d < dim; ++d) {
                on = on && (cg.g[d] >= 2 ? code % 3 != 0 : code % 3 == 0);
                code /= 3;
            }
        }
        if (on) {
            act.idx[cls] = (int8_t)act.n;
            act.cls[act.n++] = (int8_t)cls;
        }
    }
    return act;
}

static int64_t gdsw_ldp(const fedd_ctx* c) { return (c->n_rows + 15) & ~(int64_t)15; }

static int gdsw_restrict(fedd_ctx* c, const double* d_rv, double* d_r0) {
    const int dim = c->dim, dofs = c->dofs, nns = c->co_nns;
    const CoarseGeom cg = c->co_geom;
    const GdAct act = gdsw_active(cg);
    const int nsa = act.n * nns;
    const int64_t ncell = c->co_ncell, n_ent = c->co_nlat;
    if (nsa > 0)   // (RGDSW on a lattice without coarse nodes has no functions: r0 = 0 below)
        hipLaunchKernelGGL(k_gd_restrict_cells, dim3((unsigned)ncell, (unsigned)((nsa + 7) / 8)), dim3(256), 0, c->stream,
                       (const int32_t*)c->d_co_cell_ptr.p, (const int32_t*)c->d_co_val[c->co_sorted].p, dofs, nsa, gdsw_ldp(c),
                       (const double*)c->d_gd_phi.p, d_rv, c->d_co_part.p);
    const dim3 ge((unsigned)((n_ent * nns + 255) / 256)), blk(256);
    if (dim == 3) hipLaunchKernelGGL(k_gd_restrict_ent<3>, ge, blk, 0, c->stream, cg, act, nns, n_ent, (const double*)c->d_co_part.p, d_r0);
    else hipLaunchKernelGGL(k_gd_restrict_ent<2>, ge, blk, 0, c->stream, cg, act, nns, n_ent, (const double*)c->d_co_part.p, d_r0);
    return 0;
}

template <bool ADD>
static int gdsw_prolong(fedd_ctx* c, const double* d_z0, double* d_z) {
    const CoarseGeom cg = c->co_geom;
    const GdAct act = gdsw_active(cg);
    const dim3 gr((unsigned)((c->n_rows + 255) / 256)), blk(256);
    if (c->dim == 3)
        hipLaunchKernelGGL((k_gd_prolong<3, ADD>), gr, blk, 0, c->stream, cg, act, (const int32_t*)c->d_gd_ent.p, c->dofs, c->co_nns, c->n_rows,
                           gdsw_ldp(c), (const double*)c->d_gd_phi.p, (const double*)c->d_co_mask.p, d_z0, d_z);
    else
        hipLaunchKernelGGL((k_gd_prolong<2, ADD>), gr, blk, 0, c->stream, cg, act, (const int32_t*)c->d_gd_ent.p, c->dofs, c->co_nns, c->n_rows,
                           gdsw_ldp(c), (const double*)c->d_gd_phi.p, (const double*)c->d_co_mask.p, d_z0, d_z);
    return 0;
}

static int gdsw_setup(fedd_ctx* c) {
    ScopedTimer timer(c, FEDD_T_COARSE_SETUP);
    const int dim = c->dim, dofs = c->dofs;
    const int32_t n_own = (int32_t)c->n_own;
    const int64_t n_rows = c->n_rows;
    FEDD_CHECK(n_own > 0 && dofs >= 1 && dofs <= MAX_DOFS, "GDSW setup: %d owned nodes, %d dofs per node", n_own, dofs);
    FEDD_CHECK(c->have_adj, "GDSW setup: no node -> element adjacency (fedd_pattern_build first)");
    FEDD_CHECK(n_rows == (int64_t)n_own * dofs, "GDSW setup: node-interleaved systems only");
    double lo[3], hi[3], n_global = 0.0;
    FEDD_TRY(global_box(c, n_own, lo, hi, &n_global));
    // ---- the coarse decomposition: cells of a regular lattice, g_d = max(1, floor(L_d / H + 0.5)) ----
    // default: one cell per 1000 nodes, at most as many as the dense coarse solver takes ((2 g - 1)^dim * dofs coarse dofs: 10^3
    // cells for scalar problems in 3D, 7^3 for three dofs per node).  cfg 5's share (94^3-cell elasticity), setup + solve:
    // 5^3 cells 1199 + 294 ms (224 iterations), 6^3 900 + 246 (190), 7^3 869 + 215 (162): smaller interiors take fewer
    // extension iterations, and the Galerkin product no longer costs a sweep per coarse column
    const bool reduced = c->co_kind == FEDD_COARSE_RGDSW;
    // functions per entity: the translations, and with option "gdsw_rotations" on a vector problem (dofs = dim) the rotations
    const bool rot = c->gdsw_rot && dofs == dim && dim >= 2;
    const int nns = rot ? dofs + (dim == 3 ? 3 : 1) : dofs;
    c->co_nns = nns;
    double target = c->co_cells_target;
    // RGDSW has (g - 1)^dim * dofs coarse dofs only: one cell per 400 nodes, as many as the dense coarse solver takes
    // (vector problems in 3D: 13^3 cells = 5184 coarse dofs; cfg 5's share: 123 outer iterations against 186 with 9^3 cells,
    // setup 1.6 against 1.7 s -- profiles/r02_gdsw_tol_sweep.txt)
    if (!(target > 0)) {
        if (reduced) {
            const int gmax = (int)std::floor(std::pow((double)COARSE_MAX_DOFS / nns, 1.0 / dim)) + 1;
            target = std::min(std::pow((double)std::min(gmax, 20), (double)dim), std::max(1.0, std::floor(n_global / 400.0)));
        } else {
            const int gmax = std::max(1, ((int)std::floor(std::pow((double)COARSE_MAX_DOFS / nns, 1.0 / dim)) + 1) / 2);
            target = std::min(std::pow((double)gmax, (double)dim), std::max(1.0, std::floor(n_global / 1000.0)));
        }
    }
    CoarseGeom cg;
    cg.dim = dim;
    cg.reduced = reduced ? 1 : 0;
    double V = 1.0;
    for (int d = 0; d < dim; ++d) V *= (hi[d] - lo[d] > 0 ? hi[d] - lo[d] : 1.0);
    const double H = std::pow(V / target, 1.0 / dim);
    int64_t ncell = 1, n_ent = 1;
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
        n_ent *= reduced ? (g >= 2 ? g - 1 : 1) : 2 * g - 1;
    }
    const int64_t n0 = n_ent * nns;
    FEDD_CHECK(n0 <= COARSE_MAX_DOFS, "GDSW setup: %lld coarse dofs ((2g - 1)^dim entities, RGDSW: (g - 1)^dim coarse nodes, x %d), the dense coarse solver takes at "
               "most %d; lower fedd_schwarz_set_coarse (now %g cells)", (long long)n0, nns, COARSE_MAX_DOFS, target);
    const int64_t ld = (n0 + NB - 1) / NB * NB;
    c->co_geom = cg;
    c->co_ncell = ncell;
    c->co_nlat = n_ent;     // (the coarse index space: entities of the doubled lattice)
    c->co_n0 = n0;
    c->co_ld = ld;
    const GdAct act = gdsw_active(cg);
    const int nsd = act.n * nns;            // active slots (compact)
    const int64_t ldp = gdsw_ldp(c);
    const dim3 blk(256), gn((n_own + 255) / 256), gr((unsigned)((n_rows + 255) / 256));
    // ---- entity and home cell of every owned node; nodes grouped by home cell (stable radix split) ----
    for (int q = 0; q < 2; ++q) {
        FEDD_TRY(c->d_co_key[q].ensure((size_t)n_own));
        FEDD_TRY(c->d_co_val[q].ensure((size_t)n_own));
    }
    FEDD_TRY(c->d_co_cell_ptr.ensure((size_t)ncell + 1));
    FEDD_TRY(c->d_itmp0.ensure((size_t)n_own));
    FEDD_TRY(c->d_gd_ent.ensure((size_t)n_own));
    if (dim == 3)
        hipLaunchKernelGGL(k_gd_node_entity<3>, gn, blk, 0, c->stream, cg, (const int32_t*)c->d_conn.p, c->nen, (const double*)c->d_xyz.p,
                           (const int32_t*)c->d_n2e_ptr.p, (const int32_t*)c->d_n2e.p, n_own, c->d_gd_ent.p, c->d_co_key[0].p, c->d_co_val[0].p);
    else
        hipLaunchKernelGGL(k_gd_node_entity<2>, gn, blk, 0, c->stream, cg, (const int32_t*)c->d_conn.p, c->nen, (const double*)c->d_xyz.p,
                           (const int32_t*)c->d_n2e_ptr.p, (const int32_t*)c->d_n2e.p, n_own, c->d_gd_ent.p, c->d_co_key[0].p, c->d_co_val[0].p);
    int cur = 0;
    FEDD_TRY(sort_nodes_by_cell(c, n_own, ncell, &cur));
    c->co_sorted = cur;
    hipLaunchKernelGGL(k_cell_bounds, gn, blk, 0, c->stream, (const int32_t*)c->d_co_key[cur].p, n_own, (int32_t)ncell,
                       c->d_co_cell_ptr.p);
    // ---- Dirichlet mask over the column space ----
    FEDD_TRY(c->d_co_mask.ensure((size_t)c->n_cols));
    FEDD_TRY(c->d_flags.ensure(16));
    int32_t* d_bad = c->d_flags.p + 3;
    FEDD_HIP(hipMemsetAsync(d_bad, 0, 3 * sizeof(int32_t), c->stream));
    hipLaunchKernelGGL(k_mask, gr, blk, 0, c->stream, (const int32_t*)c->d_isdir.p, n_rows, c->d_co_mask.p, d_bad + 2);
    if (c->n_cols != c->n_rows || !c->halo.peers.empty()) FEDD_TRY(halo_import(c, c->d_co_mask.p, dofs));
    // ---- Phi_Gamma, then the harmonic extensions: one constrained solve per (class, component) ----
    FEDD_TRY(c->d_gd_phi.ensure((size_t)ldp * std::max(nsd, 1)));
    FEDD_TRY(c->d_gd_imask.ensure((size_t)n_rows));
    const int64_t nc = (std::max<int64_t>(n_rows, c->n_cols) + 15) & ~(int64_t)15;
    FEDD_TRY(c->d_gd_tmp.ensure((size_t)nc + 3 * (size_t)n_rows));
    double* v = c->d_gd_tmp.p;              // [nc] column-space vector (ghost tail)
    double* w = v + nc;
    double* b = w + n_rows;
    double* x = b + n_rows;
    const int32_t* d_keep = nullptr;
    if (rot) {
        // which functions of every entity are independent on its free interface dofs: Gram matrices (summed over the ranks), selection
        double hmax2 = 0.0;
        for (int d = 0; d < dim; ++d) hmax2 = std::max(hmax2, (cg.L[d] / cg.g[d]) * (cg.L[d] / cg.g[d]));
        FEDD_CHECK(n_ent * GD_NG < ((int64_t)1 << 31), "GDSW setup: %lld entities", (long long)n_ent);
        FEDD_TRY(c->d_gd_gram.ensure((size_t)n_ent * GD_NG));
        FEDD_TRY(c->d_gd_keep.ensure((size_t)n_ent));
        FEDD_HIP(hipMemsetAsync(c->d_gd_gram.p, 0, (size_t)n_ent * GD_NG * sizeof(double), c->stream));
        if (dim == 3) hipLaunchKernelGGL(k_gd_gram<3>, gn, blk, 0, c->stream, cg, act, (const int32_t*)c->d_gd_ent.p, dofs, nns, n_own, (const double*)c->d_xyz.p, (const double*)c->d_co_mask.p, c->d_gd_gram.p);
        else hipLaunchKernelGGL(k_gd_gram<2>, gn, blk, 0, c->stream, cg, act, (const int32_t*)c->d_gd_ent.p, dofs, nns, n_own, (const double*)c->d_xyz.p, (const double*)c->d_co_mask.p, c->d_gd_gram.p);
        if (c->nranks > 1) FEDD_TRY(allreduce_sum(c, c->d_gd_gram.p, (int)(n_ent * GD_NG)));
        hipLaunchKernelGGL(k_gd_select, dim3((unsigned)((n_ent + 255) / 256)), blk, 0, c->stream, (const double*)c->d_gd_gram.p, n_ent, dofs, nns, hmax2, c->d_gd_keep.p);
        d_keep = c->d_gd_keep.p;
    }
    if (dim == 3) hipLaunchKernelGGL(k_gd_phi_init<3>, gr, blk, 0, c->stream, cg, act, (const int32_t*)c->d_gd_ent.p, dofs, nns, n_rows, ldp, (const double*)c->d_xyz.p, d_keep, (const double*)c->d_co_mask.p, c->d_gd_phi.p, c->d_gd_imask.p);
    else hipLaunchKernelGGL(k_gd_phi_init<2>, gr, blk, 0, c->stream, cg, act, (const int32_t*)c->d_gd_ent.p, dofs, nns, n_rows, ldp, (const double*)c->d_xyz.p, d_keep, (const double*)c->d_co_mask.p, c->d_gd_phi.p, c->d_gd_imask.p);
    FEDD_TRY(c->d_co_part.ensure((size_t)ncell * std::max(nsd, 1)));
    FEDD_TRY(c->d_co_r0.ensure(std::max<size_t>((size_t)ld, c->d_co_r0.cap)));
    FEDD_TRY(c->d_co_z0.ensure((size_t)ld));
    int its_max = 0;
    double rel_max = 0.0;
    // tolerance of the extension solves (option "gdsw_tol"; 0 = by coarse space: GDSW 1e-4 -- below that the outer count is that of
    // exact extensions, above it grows: 163 -> 195 at 1e-3 --, RGDSW 1e-3: its outer count barely moves (123 -> 125 at cfg 5's
    // share, 62 -> 68 with the rotations) while the setup loses a third: profiles/r03_gdsw_tol_sweep_stacked.txt,
    // r04_cfg5_extension_tolerance.txt)
    const double ext_tol = c->gdsw_tol > 0.0 ? c->gdsw_tol : (reduced ? 1e-3 : 1e-4);
    // all columns solve with the same constrained operator: MULTI_NR of them at a time as one stacked system (multi.hip) -- the
    // matrix and the local inverses are read once per sweep for all of them; the stacked GMRES minimises the residual of the
    // whole block with one polynomial, columns that are zero stay zero
    const bool stacked = c->gdsw_block && c->gmres_kind == 2 && nsd > 1 && multi_rhs_ok(c);
    if (stacked) {
        const int nbatch = (nsd + MULTI_NR - 1) / MULTI_NR, nb_max = (nsd + nbatch - 1) / nbatch;
        const int64_t ns = n_rows * MULTI_NR, ncs = ((std::max<int64_t>(n_rows, c->n_cols) * MULTI_NR) + 15) & ~(int64_t)15;
        FEDD_TRY(c->d_gd_stack.ensure((size_t)ncs + 3 * (size_t)ns));
        double* Vs = c->d_gd_stack.p;       // [ncs] with the ghost rows behind the owned ones
        double* Ws = Vs + ncs;
        double* Bs = Ws + ns;
        double* Xs = Bs + ns;
        const dim3 gs((unsigned)((ns + 255) / 256));
        // the Krylov basis of a stacked solve: restart length by the memory it may take (16 GB), at least 24
        const int restart_s = (int)std::max<int64_t>(24, std::min<int64_t>(100, (int64_t)(16.0e9 / (8.0 * (double)ns)) - 1));
        for (int slot0 = 0; slot0 < nsd; slot0 += nb_max) {
            const int nb = std::min(nb_max, nsd - slot0);
            hipLaunchKernelGGL(k_gd_gamma_cols, gs, blk, 0, c->stream, (const double*)c->d_gd_phi.p, ldp, slot0, nb,
                               (const double*)c->d_gd_imask.p, n_rows, (const double*)c->d_co_mask.p, Vs);
            FEDD_TRY(spmm_owned(c, Vs, Ws, nullptr, nullptr));
            hipLaunchKernelGGL(k_gd_rhs_cols, gs, blk, 0, c->stream, (const double*)c->d_gd_imask.p, (const double*)Ws, n_rows, Bs);
            // columns to unit norm (see k_cols_sumsq): the tolerance then holds per column
            const int ncsb = (int)((n_rows + CS_ROWS - 1) / CS_ROWS);
            FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>((size_t)ncsb * MULTI_NR + 4 * MULTI_NR, c->d_dtmp0.cap)));
            double* csf = c->d_dtmp0.p;                         // [0, 16) 1 / norm, [16, 32) norm
            double* cspart = csf + 4 * MULTI_NR;
            hipLaunchKernelGGL(k_cols_sumsq, dim3((unsigned)ncsb), blk, 0, c->stream, (const double*)Bs, n_rows, cspart);
            hipLaunchKernelGGL(k_cols_sum_parts, dim3(1), dim3(64), 0, c->stream, (const double*)cspart, ncsb, csf);
            FEDD_TRY(allreduce_sum(c, csf, MULTI_NR));
            hipLaunchKernelGGL(k_cols_scale_setup, dim3(1), dim3(64), 0, c->stream, csf);
            hipLaunchKernelGGL(k_cols_scale, gs, blk, 0, c->stream, Bs, ns, (const double*)csf);
            int its = 0;
            double rel = 0.0;
            c->gm_mask = c->d_gd_imask.p;
            c->gm_nr = MULTI_NR;
            const bool timing = c->timing;
            c->timing = false;
            const int rc = gmres_solve(c, Bs, Xs, ext_tol, 1000, restart_s, 1, &its, &rel);
            c->timing = timing;
            c->gm_mask = nullptr;
            c->gm_nr = 0;
            if (rc) return rc;
            hipLaunchKernelGGL(k_cols_scale, gs, blk, 0, c->stream, Xs, ns, (const double*)(csf + MULTI_NR));
            its_max = std::max(its_max, its);
            rel_max = std::max(rel_max, rel);
            hipLaunchKernelGGL(k_gd_store_cols, dim3((unsigned)((n_rows + 15) / 16)), blk, 0, c->stream, (const double*)c->d_gd_imask.p,
                               (const double*)Xs, n_rows, c->d_gd_phi.p, ldp, slot0, nb);
        }
    }
    for (int slot = 0; slot < (stacked ? 0 : nsd); ++slot) {
        FEDD_HIP(hipMemsetAsync(v, 0, (size_t)nc * sizeof(double), c->stream));
        hipLaunchKernelGGL(k_gd_gamma_col, gr, blk, 0, c->stream, (const double*)(c->d_gd_phi.p + (int64_t)slot * ldp),
                           (const double*)c->d_gd_imask.p, n_rows, (const double*)c->d_co_mask.p, v);
        FEDD_TRY(spmv_owned(c, v, w, true));
        hipLaunchKernelGGL(k_gd_rhs, gr, blk, 0, c->stream, (const double*)c->d_gd_imask.p, (const double*)w, n_rows, b);
        int its = 0;
        double rel = 0.0;
        c->gm_mask = c->d_gd_imask.p;
        // the launches of these inner solves belong to the coarse setup (whose timer is open), not to the per-iteration
        // classes of the outer solve: their timers are suspended (the tables double-counted them)
        const bool timing = c->timing;
        c->timing = false;
        const int rc = gmres_solve(c, b, x, ext_tol, 1000, 100, 1, &its, &rel);
        c->timing = timing;
        c->gm_mask = nullptr;
        if (rc) return rc;
        its_max = std::max(its_max, its);
        rel_max = std::max(rel_max, rel);
        hipLaunchKernelGGL(k_gd_store, gr, blk, 0, c->stream, (const double*)c->d_gd_imask.p, (const double*)x, n_rows,
                           c->d_gd_phi.p + (int64_t)slot * ldp);
    }
    c->gdsw_ext_its = its_max;
    c->gdsw_ext_rel = rel_max;
    FEDD_CHECK(rel_max <= std::max(1e3 * ext_tol, 1e-6), "GDSW setup: an extension solve stopped at relative residual %.2e "
               "(tolerance %.1e)", rel_max, ext_tol);
    // ---- K0 = Phi^T A Phi, a colour (entity coordinates modulo 5) and a component at a time ----
    FEDD_TRY(c->d_co_K.ensure((size_t)ld * ld));
    FEDD_HIP(hipMemsetAsync(c->d_co_K.p, 0, (size_t)ld * ld * sizeof(double), c->stream));
    const dim3 ge((unsigned)((n0 + 255) / 256));
    int ncol[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d) ncol[d] = std::min(gd_period(cg), 2 * cg.g[d] - 1);
    // RGDSW: coarse nodes have odd coordinates in every direction with >= 2 cells (and 0 in the others): only those residues
    auto colour_used = [&](int d, int cc) { return !reduced || (cg.g[d] >= 2 ? (cc & 1) == 1 : cc == 0); };
    std::vector<GdCol> colours;
    for (int c2 = 0; c2 < ncol[2]; ++c2)
        for (int c1 = 0; c1 < ncol[1]; ++c1)
            for (int c0 = 0; c0 < ncol[0]; ++c0) {
                if (!(colour_used(0, c0) && (dim < 2 || colour_used(1, c1)) && (dim < 3 || colour_used(2, c2)))) continue;
                GdCol col;
                col.c[0] = c0; col.c[1] = c1; col.c[2] = c2;
                colours.push_back(col);
            }
    if (stacked) {
        // MULTI_NR (colour, component) columns per sweep: one SpMM, and Phi read once per sweep by the restriction
        double* Vs = c->d_gd_stack.p;
        double* Ws = Vs + (((std::max<int64_t>(n_rows, c->n_cols) * MULTI_NR) + 15) & ~(int64_t)15);
        // chunks per cell: about 2048 rows each, so that few large cells still spread over the device
        const int nch = (int)std::min<int64_t>(64, std::max<int64_t>(1, (n_rows / std::max<int64_t>(ncell, 1) + 2047) / 2048));
        FEDD_TRY(c->d_co_part.ensure((size_t)ncell * nch * std::max(nsd, 1) * MULTI_NR));
        FEDD_TRY(c->d_co_r0.ensure(std::max<size_t>((size_t)n0 * MULTI_NR, c->d_co_r0.cap)));
        FEDD_CHECK(n0 * MULTI_NR < ((int64_t)1 << 31), "GDSW setup: coarse space too large for the stacked Galerkin product");
        const dim3 gs((unsigned)((n_rows + 255) / 256)), ges((unsigned)((n0 * MULTI_NR + 255) / 256));
        const int64_t npairs = (int64_t)colours.size() * nns;
        for (int64_t p0 = 0; p0 < npairs; p0 += MULTI_NR) {
            GdCols cols;
            cols.nb = (int)std::min<int64_t>(MULTI_NR, npairs - p0);
            for (int j = 0; j < MULTI_NR; ++j) {
                const int64_t pj = std::min(p0 + j, npairs - 1);
                for (int d = 0; d < 3; ++d) cols.c[j][d] = (int8_t)colours[(size_t)(pj / nns)].c[d];
                cols.k[j] = (int8_t)(pj % nns);
            }
            if (dim == 3) hipLaunchKernelGGL(k_gd_prolong_colours<3>, gs, blk, 0, c->stream, cg, act, (const int32_t*)c->d_gd_ent.p, dofs, nns, n_rows, ldp, (const double*)c->d_gd_phi.p, (const double*)c->d_co_mask.p, cols, Vs);
            else hipLaunchKernelGGL(k_gd_prolong_colours<2>, gs, blk, 0, c->stream, cg, act, (const int32_t*)c->d_gd_ent.p, dofs, nns, n_rows, ldp, (const double*)c->d_gd_phi.p, (const double*)c->d_co_mask.p, cols, Vs);
            FEDD_TRY(spmm_owned(c, Vs, Ws, nullptr, nullptr));
            hipLaunchKernelGGL(k_gd_restrict_cells_cols, dim3((unsigned)ncell, (unsigned)((nsd + 15) / 16), (unsigned)nch), blk, 0, c->stream,
                               (const int32_t*)c->d_co_cell_ptr.p, (const int32_t*)c->d_co_val[c->co_sorted].p, dofs, nsd, nch, ldp,
                               (const double*)c->d_gd_phi.p, (const double*)Ws, c->d_co_part.p);
            if (dim == 3) hipLaunchKernelGGL(k_gd_restrict_ent_cols<3>, ges, blk, 0, c->stream, cg, act, nns, n_ent, nch, (const double*)c->d_co_part.p, c->d_co_r0.p);
            else hipLaunchKernelGGL(k_gd_restrict_ent_cols<2>, ges, blk, 0, c->stream, cg, act, nns, n_ent, nch, (const double*)c->d_co_part.p, c->d_co_r0.p);
            if (c->nranks > 1) FEDD_TRY(allreduce_sum(c, c->d_co_r0.p, (int)(n0 * MULTI_NR)));
            if (dim == 3) hipLaunchKernelGGL(k_gd_scatter_cols<3>, ges, blk, 0, c->stream, cg, nns, n_ent, cols, (const double*)c->d_co_r0.p, c->d_co_K.p, ld);
            else hipLaunchKernelGGL(k_gd_scatter_cols<2>, ges, blk, 0, c->stream, cg, nns, n_ent, cols, (const double*)c->d_co_r0.p, c->d_co_K.p, ld);
        }
        // (the per-column restriction of the apply takes its own layout of the partial sums)
        FEDD_TRY(c->d_co_part.ensure((size_t)ncell * std::max(nsd, 1)));
    }
    for (size_t ci = 0; ci < (stacked ? 0 : colours.size()); ++ci)
                for (int k = 0; k < nns; ++k) {
                    const GdCol col = colours[ci];
                    if (dim == 3) hipLaunchKernelGGL(k_gd_prolong_colour<3>, gr, blk, 0, c->stream, cg, act, (const int32_t*)c->d_gd_ent.p, dofs, nns, n_rows, ldp, (const double*)c->d_gd_phi.p, (const double*)c->d_co_mask.p, k, col, v);
                    else hipLaunchKernelGGL(k_gd_prolong_colour<2>, gr, blk, 0, c->stream, cg, act, (const int32_t*)c->d_gd_ent.p, dofs, nns, n_rows, ldp, (const double*)c->d_gd_phi.p, (const double*)c->d_co_mask.p, k, col, v);
                    FEDD_TRY(spmv_owned(c, v, w, true));
                    FEDD_TRY(gdsw_restrict(c, w, c->d_co_r0.p));
                    if (c->nranks > 1) FEDD_TRY(allreduce_sum(c, c->d_co_r0.p, (int)n0));
                    if (dim == 3) hipLaunchKernelGGL(k_gd_scatter_col<3>, ge, blk, 0, c->stream, cg, nns, n_ent, k, col, (const double*)c->d_co_r0.p, c->d_co_K.p, ld);
                    else hipLaunchKernelGGL(k_gd_scatter_col<2>, ge, blk, 0, c->stream, cg, nns, n_ent, k, col, (const double*)c->d_co_r0.p, c->d_co_K.p, ld);
                }
    hipLaunchKernelGGL(k_fix_diag, dim3((unsigned)((ld + 3) / 4)), blk, 0, c->stream, c->d_co_K.p, ld, n0);
    FEDD_TRY(dense_invert_batched(c, c->d_co_K.p, ld, 1, ld * ld, nullptr, (int)(ld / NB), 1, d_bad + 1));
    int32_t bad[3] = {0, 0, 0};
    FEDD_HIP(hipMemcpyAsync(bad, d_bad, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    if (c->nranks > 1) {
        double hb = (double)bad[1];
        FEDD_HIP(hipMemcpyAsync(c->d_co_r0.p, &hb, sizeof(hb), hipMemcpyHostToDevice, c->stream));
        FEDD_TRY(allreduce_sum(c, c->d_co_r0.p, 1));
        FEDD_HIP(hipMemcpyAsync(&hb, c->d_co_r0.p, sizeof(hb), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        bad[1] = hb != 0.0;
    }
    FEDD_CHECK(!bad[1], "GDSW setup: K0 is not positive definite (the interior extensions were not solved well enough, or the "
                        "matrix is not SPD on its free dofs)");
    FEDD_HIP(hipGetLastError());
    c->have_coarse = true;
    return 0;
}

static int gdsw_apply_add(fedd_ctx* c, const double* d_r_owned, double* d_z_owned) {
    ScopedTimer timer(c, FEDD_T_COARSE_APPLY);
    const int64_t n0 = c->co_n0, ld = c->co_ld;
    FEDD_TRY(gdsw_restrict(c, d_r_owned, c->d_co_r0.p));
    if (c->nranks > 1) {
        timer.stop();
        FEDD_TRY(allreduce_sum(c, c->d_co_r0.p, (int)n0));
        timer.resume();
    }
    hipLaunchKernelGGL(k_dense_mv, dim3((unsigned)((n0 + 3) / 4)), dim3(256), 0, c->stream, (const double*)c->d_co_K.p, ld, n0,
                       (const double*)c->d_co_r0.p, c->d_co_z0.p);
    FEDD_TRY(gdsw_prolong<true>(c, c->d_co_z0.p, d_z_owned));
    FEDD_HIP(hipGetLastError());
    return 0;
}

int coarse_setup(fedd_ctx* c) {
    if (c->co_kind == FEDD_COARSE_GDSW || c->co_kind == FEDD_COARSE_RGDSW) return gdsw_setup(c);
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
    FEDD_TRY(sort_nodes_by_cell(c, n_own, ncell, &cur));
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
    if (c->co_kind == FEDD_COARSE_GDSW || c->co_kind == FEDD_COARSE_RGDSW) return gdsw_apply_add(c, d_r_owned, d_z_owned);
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
