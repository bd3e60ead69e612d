// Numeric FE assembly on the device, gather formulation (no atomics, bitwise reproducible):
// one lane per owned dof row walks the (element, local index) list of its node, evaluates the
// row of each incident element's local matrix and adds it into an LDS-resident copy of the CSR
// row; the row is written to HBM once.
//
// Arithmetic follows (not copies) the reference's element loops:
//   FE::assemblyLaplace          feddlib/core/FE/FE_def.hpp:604-667
//   FE::assemblyLaplaceVecField  feddlib/core/FE/FE_def.hpp:670-734
//   FE::assemblyMass             feddlib/core/FE/FE_def.hpp:454-524
//   FE::assemblyLinElasXDim      feddlib/core/FE/FE_def.hpp:2739-3040 (epsilonTensor :4931-4944)
//   FE::assemblyRHS              feddlib/core/FE/FE_def.hpp:4694-4766
//   FE::buildTransformation      feddlib/core/FE/FE_def.hpp:5342-5357
//   SmallMatrix::computeInverse  feddlib/core/General/SmallMatrix.hpp:306-357
//   FE::applyBTinv               feddlib/core/FE/FE_def.hpp:83-96
//   BCBuilder::setSystem/setRHS  feddlib/core/General/BCBuilder_def.hpp:589-707, 93-170
// Quadrature points/weights and reference basis values/gradients are staged in LDS once per
// workgroup.
#include "fedd_internal.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <thread>

namespace fedd {
namespace {

// F_DIV / F_DIVT: FE::assemblyDivAndDivT (feddlib/core/FE/FE_def.hpp:1932-2057), pressure = P1 on
// the element's vertices.  F_DIV rows = pressure nodes, columns = DIM*velocity node + d;
// F_DIVT rows = velocity dofs, columns = pressure nodes.
enum { F_LAPLACE = 0, F_MASS = 1, F_LINELAS = 2, F_DIV = 3, F_DIVT = 4 };

struct AsmArgs {
    const int32_t* conn;
    const int32_t* n2e_ptr;
    const int32_t* n2e;
    const int32_t* rowptr;
    const int32_t* colind;
    const double* xyz;
    double* val;
    const double* tab;  // w[nq] | phi[nq*nen] | dphi[nq*nen*dim] | psi[nq*(dim+1)] (P1 pressure basis)
    int nq;
    int32_t n_rows;
    int dofs;
    double p0, p1;  // LINELAS: lambda, mu
    const double* ke;   // != nullptr: element matrices [E][NEN][NEN] computed beforehand by k_elem_matrix (P2 scalar forms)
    double zero_eps; // > 0: element contributions of magnitude below it are set to zero before they are added (the reference's
                     // optional setZeros_ / myeps_, FE_def.hpp:74-79, 719-721, 2002-2004, 2032-2034: vector Laplacian, B, B^T)
};
__device__ __forceinline__ double zero_small(const AsmArgs& a, double v) { return (a.zero_eps > 0.0 && fabs(v) < a.zero_eps) ? 0.0 : v; }

__device__ __forceinline__ int find_slot(const int32_t* __restrict__ cols, int n, int32_t col) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cols[mid] < col) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// affine map of a simplex: B[i][j] = x_{j+1}[i] - x_0[i]; returns det, fills Binv = adj(B)/det
template <int DIM>
__device__ __forceinline__ double affine(const double (&X)[DIM + 1][DIM], double (&Binv)[DIM][DIM]) {
    double B[DIM][DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j)
#pragma unroll
        for (int i = 0; i < DIM; ++i) B[i][j] = X[j + 1][i] - X[0][i];
    if constexpr (DIM == 2) {
        // one f64 division (the reference divides each adjugate entry by det; multiplying by the
        // correctly rounded reciprocal differs by <= 1 ulp per entry, far inside the 1e-10 bar)
        const double det = B[0][0] * B[1][1] - B[1][0] * B[0][1];
        const double rdet = 1.0 / det;
        Binv[0][0] = B[1][1] * rdet;
        Binv[0][1] = (-B[0][1]) * rdet;
        Binv[1][0] = (-B[1][0]) * rdet;
        Binv[1][1] = B[0][0] * rdet;
        return det;
    } else {
        const double det = B[0][0] * B[1][1] * B[2][2] + B[0][1] * B[1][2] * B[2][0] + B[0][2] * B[1][0] * B[2][1] -
                           B[2][0] * B[1][1] * B[0][2] - B[2][1] * B[1][2] * B[0][0] - B[2][2] * B[1][0] * B[0][1];
        const double rdet = 1.0 / det;
        Binv[0][0] = (B[1][1] * B[2][2] - B[1][2] * B[2][1]) * rdet;
        Binv[0][1] = (B[0][2] * B[2][1] - B[0][1] * B[2][2]) * rdet;
        Binv[0][2] = (B[0][1] * B[1][2] - B[0][2] * B[1][1]) * rdet;
        Binv[1][0] = (B[1][2] * B[2][0] - B[1][0] * B[2][2]) * rdet;
        Binv[1][1] = (B[0][0] * B[2][2] - B[0][2] * B[2][0]) * rdet;
        Binv[1][2] = (B[0][2] * B[1][0] - B[0][0] * B[1][2]) * rdet;
        Binv[2][0] = (B[1][0] * B[2][1] - B[1][1] * B[2][0]) * rdet;
        Binv[2][1] = (B[0][1] * B[2][0] - B[0][0] * B[2][1]) * rdet;
        Binv[2][2] = (B[0][0] * B[1][1] - B[0][1] * B[1][0]) * rdet;
        return det;
    }
}

template <int DIM>
__device__ __forceinline__ double affine_det(const double (&X)[DIM + 1][DIM]) {
    double B[DIM][DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j)
#pragma unroll
        for (int i = 0; i < DIM; ++i) B[i][j] = X[j + 1][i] - X[0][i];
    if constexpr (DIM == 2) {
        return B[0][0] * B[1][1] - B[1][0] * B[0][1];
    } else {
        return B[0][0] * B[1][1] * B[2][2] + B[0][1] * B[1][2] * B[2][0] + B[0][2] * B[1][0] * B[2][1] -
               B[2][0] * B[1][1] * B[0][2] - B[2][1] * B[1][2] * B[0][0] - B[2][2] * B[1][0] * B[0][1];
    }
}

// transformed gradient of basis function i at quadrature point q: g[d] = sum_d2 dphi[q][i][d2] Binv[d2][d]
template <int DIM, int NEN>
__device__ __forceinline__ void grad_t(const double* __restrict__ s_dphi, int q, int i, const double (&Binv)[DIM][DIM],
                                       double (&g)[DIM]) {
    const double* dp = s_dphi + (q * NEN + i) * DIM;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        double s = 0.0;
#pragma unroll
        for (int d2 = 0; d2 < DIM; ++d2) s += dp[d2] * Binv[d2][d];
        g[d] = s;
    }
}

template <int DIM, int NEN, int FORM>
__global__ void k_assemble(AsmArgs a) {
    extern __shared__ double sm[];
    const int nq = a.nq;
    double* s_w = sm;
    double* s_phi = s_w + nq;
    double* s_dphi = s_phi + nq * NEN;
    double* acc = s_dphi + nq * NEN * DIM;
    const int tid = threadIdx.x;
    const int BS = blockDim.x;
    const int ntab = nq * (1 + NEN + NEN * DIM);
    for (int i = tid; i < ntab; i += BS) sm[i] = a.tab[i];
    __syncthreads();
    const int32_t row = blockIdx.x * BS + tid;
    if (row >= a.n_rows) return;
    const int dofs = a.dofs;
    const int32_t node = row / dofs;
    const int comp = row - node * dofs;
    const int32_t rs = a.rowptr[row];
    const int rn = a.rowptr[row + 1] - rs;
    const int32_t* __restrict__ cols = a.colind + rs;
    for (int s = 0; s < rn; ++s) acc[s * BS + tid] = 0.0;

    for (int32_t p = a.n2e_ptr[node]; p < a.n2e_ptr[node + 1]; ++p) {
        const int32_t idx = a.n2e[p];
        const int32_t e = idx / NEN;
        const int li = idx - e * NEN;
        int32_t nd[NEN];
#pragma unroll
        for (int j = 0; j < NEN; ++j) nd[j] = a.conn[(int64_t)e * NEN + j];
        double X[DIM + 1][DIM];
#pragma unroll
        for (int v = 0; v <= DIM; ++v)
#pragma unroll
            for (int d = 0; d < DIM; ++d) X[v][d] = a.xyz[(int64_t)nd[v] * DIM + d];

        if constexpr (FORM == F_MASS) {
            const double absdet = fabs(affine_det<DIM>(X));
#pragma unroll
            for (int j = 0; j < NEN; ++j) {
                double v = 0.0;
                for (int q = 0; q < nq; ++q) v += s_w[q] * s_phi[q * NEN + li] * s_phi[q * NEN + j];
                v *= absdet;
                v -= a.p0 * absdet * a.p1;   // assemblyBDStabilization (FE_def.hpp:2151-2220): p0 = |reference element|, p1 = its scale; mass: 0
                const int slot = find_slot(cols, rn, nd[j] * dofs + comp);
                acc[slot * BS + tid] += v;
            }
        } else {
            double Binv[DIM][DIM];
            const double absdet = fabs(affine<DIM>(X, Binv));
            if constexpr (FORM == F_LAPLACE) {
#pragma unroll
                for (int j = 0; j < NEN; ++j) {
                    double v = 0.0;
                    for (int q = 0; q < nq; ++q) {
                        double gi[DIM], gj[DIM];
                        grad_t<DIM, NEN>(s_dphi, q, li, Binv, gi);
                        grad_t<DIM, NEN>(s_dphi, q, j, Binv, gj);
#pragma unroll
                        for (int d = 0; d < DIM; ++d) v += s_w[q] * gi[d] * gj[d];
                    }
                    v *= absdet;
                    const int slot = find_slot(cols, rn, nd[j] * dofs + comp);
                    acc[slot * BS + tid] += v;
                }
            } else {  // F_LINELAS, row (node, comp): full dofs x dofs coupling, dofs == DIM
                const double lam = a.p0, mu = a.p1;
#pragma unroll
                for (int j = 0; j < NEN; ++j) {
                    double vb[DIM];
#pragma unroll
                    for (int b = 0; b < DIM; ++b) vb[b] = 0.0;
                    for (int q = 0; q < nq; ++q) {
                        double gi[DIM], gj[DIM];
                        grad_t<DIM, NEN>(s_dphi, q, li, Binv, gi);
                        grad_t<DIM, NEN>(s_dphi, q, j, Binv, gj);
                        double dot = 0.0;
#pragma unroll
                        for (int d = 0; d < DIM; ++d) dot += gi[d] * gj[d];
                        double gia = 0.0, gja = 0.0;
#pragma unroll
                        for (int d = 0; d < DIM; ++d) {
                            gia = d == comp ? gi[d] : gia;
                            gja = d == comp ? gj[d] : gja;
                        }
#pragma unroll
                        for (int b = 0; b < DIM; ++b) {
                            // 2 mu eps_i:eps_j + lam tr(eps_i) tr(eps_j) with eps from epsilonTensor
                            const double e = mu * ((b == comp ? dot : 0.0) + gi[b] * gja) + lam * gia * gj[b];
                            vb[b] += s_w[q] * e;
                        }
                    }
                    const int slot0 = find_slot(cols, rn, nd[j] * dofs);
#pragma unroll
                    for (int b = 0; b < DIM; ++b) acc[(slot0 + b) * BS + tid] += absdet * vb[b];
                }
            }
        }
    }
    for (int s = 0; s < rn; ++s) a.val[rs + s] = acc[s * BS + tid];
}

// ---------------------------------------------------------------------------------------------
// Pair-parallel variant (default).  A workgroup owns R consecutive dof rows.
//   phase 1: one lane per (row, incident element) pair evaluates that row of the element matrix
//            and parks (column id, value) in LDS -- all global-memory latency (adjacency, element
//            nodes, coordinates) is overlapped across ~24x more lanes than rows;
//   phase 2: `tpr` lanes per row, lane s owns CSR slot s: it sweeps the row's parked
//            contributions in their fixed order and adds those whose column id is its own
//            (LDS broadcast reads, no search, no atomics), then the workgroup's CSR range
//            is written as one contiguous coalesced stream.
// Summation order is fixed by the sorted adjacency list => bitwise reproducible.
// ---------------------------------------------------------------------------------------------
template <int DIM, int NEN, int FORM>
struct PairCfg {
    // contributions per (row, element) pair
    static constexpr int CPP = (FORM == F_LINELAS || FORM == F_DIV) ? NEN * DIM : (FORM == F_DIVT ? DIM + 1 : NEN);
};

// row `li` (component `comp`) of the element matrix of an element with nodes nd and vertex coordinates X:
// CPP (column id, value) contributions
template <int DIM, int NEN, int FORM>
__device__ __forceinline__ void compute_pair(const AsmArgs& a, const double* __restrict__ s_w,
                                             const double* __restrict__ s_phi, const double* __restrict__ s_dphi, int nq,
                                             const int32_t (&nd)[NEN], const double (&X)[DIM + 1][DIM], int li, int comp,
                                             int dofs, int32_t (&cols)[PairCfg<DIM, NEN, FORM>::CPP],
                                             double (&vals)[PairCfg<DIM, NEN, FORM>::CPP]) {
    if constexpr (FORM == F_MASS) {
        const double absdet = fabs(affine_det<DIM>(X));
#pragma unroll
        for (int j = 0; j < NEN; ++j) {
            double v = 0.0;
            for (int q = 0; q < nq; ++q) v += s_w[q] * s_phi[q * NEN + li] * s_phi[q * NEN + j];
            cols[j] = nd[j] * dofs + comp;
            // assemblyBDStabilization (FE_def.hpp:2204-2206): value *= absDetB; value -= refElementSize * absDetB * refElementScale
            // (p0 = p1 = 0 for the plain mass matrix: x - 0 = x)
            vals[j] = v * absdet - a.p0 * absdet * a.p1;
        }
    } else {
        // Quadrature loop outermost, the transformed gradient of the row's basis function once per point and that
        // of each column function once per (point, column).  On P1 elements the gradients do not depend on the
        // point: all of them are formed once.  (The kernel is bound by f64 issue: the earlier form re-derived both
        // gradients inside the column loop, 345 f64 instructions per pair for P1 Laplace against ~130 now; the
        // expressions and their order are unchanged, so are the bits.)
        constexpr bool P1 = NEN == DIM + 1;
        double Binv[DIM][DIM];
        const double absdet = fabs(affine<DIM>(X, Binv));
        double gi[DIM], G[P1 ? NEN : 1][DIM];
        if constexpr (P1) {
            grad_t<DIM, NEN>(s_dphi, 0, li, Binv, gi);
#pragma unroll
            for (int j = 0; j < NEN; ++j) grad_t<DIM, NEN>(s_dphi, 0, j, Binv, G[j]);
        }
        if constexpr (FORM == F_LAPLACE) {
            double v[NEN];
#pragma unroll
            for (int j = 0; j < NEN; ++j) v[j] = 0.0;
            for (int q = 0; q < nq; ++q) {
                if constexpr (!P1) grad_t<DIM, NEN>(s_dphi, q, li, Binv, gi);
                double wg[DIM];
#pragma unroll
                for (int d = 0; d < DIM; ++d) wg[d] = s_w[q] * gi[d];
#pragma unroll
                for (int j = 0; j < NEN; ++j) {
                    double gj[DIM];
                    if constexpr (P1) {
#pragma unroll
                        for (int d = 0; d < DIM; ++d) gj[d] = G[j][d];
                    } else {
                        grad_t<DIM, NEN>(s_dphi, q, j, Binv, gj);
                    }
#pragma unroll
                    for (int d = 0; d < DIM; ++d) v[j] += wg[d] * gj[d];
                }
            }
#pragma unroll
            for (int j = 0; j < NEN; ++j) {
                cols[j] = nd[j] * dofs + comp;
                vals[j] = zero_small(a, v[j] * absdet);
            }
        } else if constexpr (FORM == F_DIV) {
            // row = pressure node (vertex li): B_{i,(j,d)} = |detB| sum_q w_q psi_qi dphi_qjd  (FE_def.hpp:1992-2004)
            const double* __restrict__ s_psi = s_dphi + nq * NEN * DIM;
            double vd[NEN][DIM];
#pragma unroll
            for (int j = 0; j < NEN; ++j)
#pragma unroll
                for (int d = 0; d < DIM; ++d) vd[j][d] = 0.0;
            for (int q = 0; q < nq; ++q) {
                const double wp = s_w[q] * s_psi[q * (DIM + 1) + li];
#pragma unroll
                for (int j = 0; j < NEN; ++j) {
                    double gj[DIM];
                    if constexpr (P1) {
#pragma unroll
                        for (int d = 0; d < DIM; ++d) gj[d] = G[j][d];
                    } else {
                        grad_t<DIM, NEN>(s_dphi, q, j, Binv, gj);
                    }
#pragma unroll
                    for (int d = 0; d < DIM; ++d) vd[j][d] += wp * gj[d];
                }
            }
#pragma unroll
            for (int j = 0; j < NEN; ++j)
#pragma unroll
                for (int d = 0; d < DIM; ++d) {
                    cols[j * DIM + d] = nd[j] * DIM + d;
                    vals[j * DIM + d] = zero_small(a, absdet * vd[j][d]);
                }
        } else if constexpr (FORM == F_DIVT) {
            // row = velocity dof (node li, component comp): B^T_{(i,d),j} = |detB| sum_q w_q psi_qj dphi_qid  (:2022-2046)
            const double* __restrict__ s_psi = s_dphi + nq * NEN * DIM;
            double vj[DIM + 1];
#pragma unroll
            for (int j = 0; j <= DIM; ++j) vj[j] = 0.0;
            for (int q = 0; q < nq; ++q) {
                if constexpr (!P1) grad_t<DIM, NEN>(s_dphi, q, li, Binv, gi);
                double gc = 0.0;
#pragma unroll
                for (int d = 0; d < DIM; ++d) gc = d == comp ? gi[d] : gc;
#pragma unroll
                for (int j = 0; j <= DIM; ++j) vj[j] += s_w[q] * s_psi[q * (DIM + 1) + j] * gc;
            }
#pragma unroll
            for (int j = 0; j <= DIM; ++j) {
                cols[j] = nd[j];
                vals[j] = zero_small(a, absdet * vj[j]);
            }
        } else {
            const double lam = a.p0, mu = a.p1;
            double vb[NEN][DIM];
#pragma unroll
            for (int j = 0; j < NEN; ++j)
#pragma unroll
                for (int b = 0; b < DIM; ++b) vb[j][b] = 0.0;
            for (int q = 0; q < nq; ++q) {
                if constexpr (!P1) grad_t<DIM, NEN>(s_dphi, q, li, Binv, gi);
                double gia = 0.0;
#pragma unroll
                for (int d = 0; d < DIM; ++d) gia = d == comp ? gi[d] : gia;
#pragma unroll
                for (int j = 0; j < NEN; ++j) {
                    double gj[DIM];
                    if constexpr (P1) {
#pragma unroll
                        for (int d = 0; d < DIM; ++d) gj[d] = G[j][d];
                    } else {
                        grad_t<DIM, NEN>(s_dphi, q, j, Binv, gj);
                    }
                    double dot = 0.0, gja = 0.0;
#pragma unroll
                    for (int d = 0; d < DIM; ++d) {
                        dot += gi[d] * gj[d];
                        gja = d == comp ? gj[d] : gja;
                    }
                    // 2 mu eps_i:eps_j + lam tr(eps_i) tr(eps_j) with eps from epsilonTensor (FE_def.hpp:4931-4944)
#pragma unroll
                    for (int b = 0; b < DIM; ++b)
                        vb[j][b] += s_w[q] * (mu * ((b == comp ? dot : 0.0) + gi[b] * gja) + lam * gia * gj[b]);
                }
            }
#pragma unroll
            for (int j = 0; j < NEN; ++j)
#pragma unroll
                for (int b = 0; b < DIM; ++b) {
                    cols[j * DIM + b] = nd[j] * dofs + b;
                    vals[j * DIM + b] = absdet * vb[j][b];
                }
        }
    }
}

template <int DIM, int NEN, int FORM>
__device__ __forceinline__ void eval_pair(const AsmArgs& a, const double* __restrict__ s_w,
                                          const double* __restrict__ s_phi, const double* __restrict__ s_dphi, int nq,
                                          int32_t e, int li, int comp, int dofs,
                                          int32_t (&cols)[PairCfg<DIM, NEN, FORM>::CPP],
                                          double (&vals)[PairCfg<DIM, NEN, FORM>::CPP]) {
    int32_t nd[NEN];
#pragma unroll
    for (int j = 0; j < NEN; ++j) nd[j] = a.conn[(int64_t)e * NEN + j];
    if constexpr ((FORM == F_LAPLACE || FORM == F_MASS) && NEN > DIM + 1) {
        if (a.ke) {     // row li of the element matrix k_elem_matrix left behind: NEN contiguous values
            const double* __restrict__ kr = a.ke + ((int64_t)e * NEN + li) * NEN;
#pragma unroll
            for (int j = 0; j < NEN; ++j) {
                cols[j] = nd[j] * dofs + comp;
                vals[j] = kr[j];
            }
            return;
        }
    }
    double X[DIM + 1][DIM];
#pragma unroll
    for (int v = 0; v <= DIM; ++v)
#pragma unroll
        for (int d = 0; d < DIM; ++d) X[v][d] = a.xyz[(int64_t)nd[v] * DIM + d];
    compute_pair<DIM, NEN, FORM>(a, s_w, s_phi, s_dphi, nq, nd, X, li, comp, dofs, cols, vals);
}

template <int DIM, int NEN, int FORM>
__global__ __launch_bounds__(256) void k_assemble_pairs(AsmArgs a, int R, int tpr_log2, int cap_contrib) {
    constexpr int CPP = PairCfg<DIM, NEN, FORM>::CPP;
    extern __shared__ double sm[];
    const int nq = a.nq;
    const int ntab = nq * (1 + NEN + NEN * DIM + DIM + 1);
    double* s_w = sm;
    double* s_phi = s_w + nq;
    double* s_dphi = s_phi + nq * NEN;
    double* cval = sm + ntab;
    int32_t* ccol = reinterpret_cast<int32_t*>(cval + cap_contrib);
    int32_t* off = ccol + cap_contrib;
    const int tid = threadIdx.x;
    for (int i = tid; i < ntab; i += 256) sm[i] = a.tab[i];
    const int dofs = a.dofs;
    // each XCD takes a contiguous eighth of the rows: rows of neighbouring node lines share elements, whose
    // connectivity and coordinates then meet in one L2 (PMC traffic 2.7x -> 1.75x the algorithmic bytes at 100^3 cells)
    const int32_t nwg = gridDim.x, q8 = nwg >> 3, rem8 = nwg & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int32_t wg = (xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8) + within;
    const int32_t r0 = wg * R;
    const int nrows = min(R, a.n_rows - r0);
    // exclusive prefix of the rows' pair counts: first wave, one lane per row (R <= 64), DPP scan
    if (tid < 64) {
        int deg = 0;
        if (tid < nrows) {
            const int32_t node = (r0 + tid) / dofs;
            deg = a.n2e_ptr[node + 1] - a.n2e_ptr[node];
        }
        int incl = deg;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (tid >= d) incl += t;
        }
        if (tid <= nrows) off[tid] = incl - deg;
    }
    __syncthreads();
    const int npairs = off[nrows];
    // the contributions of local row r start at (off[r]*CPP + r*PAD): the odd-ish shift keeps the
    // rows that one wave sweeps together in phase 2 on different LDS banks
    constexpr int PAD = 2;
    for (int i = tid; i < npairs; i += 256) {
        int lo = 0, hi = nrows - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (off[mid] <= i) lo = mid;
            else hi = mid - 1;
        }
        const int32_t row = r0 + lo;
        const int32_t node = row / dofs;
        const int comp = row - node * dofs;
        const int32_t idx = a.n2e[a.n2e_ptr[node] + (i - off[lo])];
        const int32_t e = idx / NEN;
        const int li = idx - e * NEN;
        int32_t cols[CPP];
        double vals[CPP];
        eval_pair<DIM, NEN, FORM>(a, s_w, s_phi, s_dphi, nq, e, li, comp, dofs, cols, vals);
        const int base = i * CPP + lo * PAD;
#pragma unroll
        for (int c = 0; c < CPP; ++c) {
            ccol[base + c] = cols[c];
            cval[base + c] = vals[c];
        }
    }
    __syncthreads();
    const int tpr = 1 << tpr_log2;
    for (int item = tid; item < (nrows << tpr_log2); item += 256) {
        const int rl = item >> tpr_log2, s0 = item & (tpr - 1);
        const int32_t rs = a.rowptr[r0 + rl];
        const int rn = a.rowptr[r0 + rl + 1] - rs;
        const int cb = off[rl] * CPP + rl * PAD, ce = off[rl + 1] * CPP + rl * PAD;
        for (int s = s0; s < rn; s += tpr) {
            const int32_t mycol = a.colind[rs + s];
            double acc = 0.0;
            int c = cb;
            // 8 contributions per trip, all 16 LDS reads issued before the first use (a
            // data-dependent read of cval would serialise two LDS latencies per contribution)
            for (; c + 8 <= ce; c += 8) {
                int32_t cc[8];
                double vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    cc[u] = ccol[c + u];
                    vv[u] = cval[c + u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += cc[u] == mycol ? vv[u] : 0.0;
            }
            for (; c < ce; ++c) {
                const int32_t cc = ccol[c];
                const double vv = cval[c];
                acc += cc == mycol ? vv : 0.0;
            }
            a.val[rs + s] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Slot-addressed variant (default, asm_kind 0).  A workgroup owns R consecutive dof rows, i.e. one
// contiguous range of the CSR arrays, and keeps an image of that range in LDS:
//   phase 0: the range's column ids -> LDS (coalesced), accumulators zeroed;
//   phase 1: one lane per (row, incident element) pair, as in the pair-parallel kernel; the global
//            loads of U pairs per lane (adjacency entry -> element nodes -> vertex coordinates) are
//            issued as three batches of independent requests; each contribution's CSR slot is found
//            by a binary search of its column id in the row's LDS-resident column list, and
//            (slot, value) is parked in LDS;
//   phase 2: one lane per row adds its parked contributions into the LDS image in their fixed order
//            (adjacency order, then local column order): no atomics, no sweep over the row's other
//            contributions -- 96 read-modify-writes per P1 row instead of 15 slots x 96 compares;
//   phase 3: the image is written to HBM as one contiguous coalesced stream.
// Summation order is fixed by the sorted adjacency list => bitwise reproducible.  Workgroups are
// remapped so that each XCD takes a contiguous eighth of the rows (rows of neighbouring node lines
// share elements: their connectivity and coordinates then meet in one L2).
// ---------------------------------------------------------------------------------------------
template <int DIM, int NEN, int FORM, int U /* pairs per lane whose loads are in flight together */>
__global__ __launch_bounds__(256) void k_assemble_slots(AsmArgs a, int R, int cap_contrib, int cap_cols, int nwg, int dbg) {
    constexpr int CPP = PairCfg<DIM, NEN, FORM>::CPP;
    constexpr int PADV = 1, PADS = 2;        // per-row shifts of the parks: the lanes of phase 2 (one per row) hit different banks
    extern __shared__ double sm[];
    const int nq = a.nq;
    const int ntab = nq * (1 + NEN + NEN * DIM + DIM + 1);
    double* s_w = sm;
    double* s_phi = s_w + nq;
    double* s_dphi = s_phi + nq * NEN;
    double* cval = sm + ntab;                                   // [cap_contrib]
    double* acc = cval + cap_contrib;                           // [cap_cols]   image of val[rs0, rs0 + ncols)
    int32_t* scol = reinterpret_cast<int32_t*>(acc + cap_cols); // [cap_cols]   image of colind[...]
    int32_t* off = scol + cap_cols;                             // [R + 1]      pair offsets of the rows
    int32_t* rbase = off + R + 1;                               // [R + 1]      row starts relative to rs0
    uint16_t* cslot = reinterpret_cast<uint16_t*>(rbase + R + 1);   // [cap_contrib] position in the image
    const int tid = threadIdx.x;
    const int32_t q8 = nwg >> 3, rem8 = nwg & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int32_t wg = (xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8) + within;
    for (int i = tid; i < ntab; i += 256) sm[i] = a.tab[i];
    const int dofs = a.dofs;
    const int32_t r0 = wg * R;
    const int nrows = min(R, a.n_rows - r0);
    const int32_t rs0 = a.rowptr[r0];
    if (tid < 64) {   // exclusive prefix of the rows' pair counts; row starts
        int deg = 0;
        if (tid < nrows) {
            const int32_t node = (r0 + tid) / dofs;
            deg = a.n2e_ptr[node + 1] - a.n2e_ptr[node];
        }
        int incl = deg;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (tid >= d) incl += t;
        }
        if (tid <= nrows) {
            off[tid] = incl - deg;
            rbase[tid] = a.rowptr[r0 + tid] - rs0;
        }
    }
    __syncthreads();
    const int npairs = off[nrows];
    const int ncols = rbase[nrows];
    for (int i = tid; i < ncols; i += 256) {
        scol[i] = a.colind[rs0 + i];
        acc[i] = 0.0;
    }
    __syncthreads();
    for (int i0 = tid; i0 < npairs && !(dbg & 2); i0 += 256 * U) {
        int lo_[U], li_[U];
        int32_t e_[U], nd[U][NEN];
        double X[U][DIM + 1][DIM];
        // batch 1: adjacency entries (row of the pair by binary search over the offsets)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(i0 + u * 256, npairs - 1);
            int lo = 0, hi = nrows - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (off[mid] <= i) lo = mid;
                else hi = mid - 1;
            }
            lo_[u] = lo;
            const int32_t node = (r0 + lo) / dofs;
            e_[u] = a.n2e[a.n2e_ptr[node] + (i - off[lo])];
        }
        // batch 2: element nodes
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int32_t e = e_[u] / NEN;
            li_[u] = e_[u] - e * NEN;
            e_[u] = e;
#pragma unroll
            for (int j = 0; j < NEN; ++j) nd[u][j] = a.conn[(int64_t)e * NEN + j];
        }
        // batch 3: vertex coordinates
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v <= DIM; ++v)
#pragma unroll
                for (int d = 0; d < DIM; ++d) X[u][v][d] = a.xyz[(int64_t)nd[u][v] * DIM + d];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256;
            if (i >= npairs) break;
            const int lo = lo_[u];
            const int comp = (r0 + lo) % dofs;
            int32_t cols[CPP];
            double vals[CPP];
            compute_pair<DIM, NEN, FORM>(a, s_w, s_phi, s_dphi, nq, nd[u], X[u], li_[u], comp, dofs, cols, vals);
            const int lb = rbase[lo], ln = rbase[lo + 1] - lb;
            const int bv = i * CPP + lo * PADV, bs = i * CPP + lo * PADS;
#pragma unroll
            for (int c = 0; c < CPP; ++c) {
                cval[bv + c] = vals[c];
                cslot[bs + c] = (uint16_t)(lb + find_slot(scol + lb, ln, cols[c]));
            }
        }
    }
    __syncthreads();
    // L lanes per row: lane t adds the contributions c = j L + t, j = 0..NEN-1 (for the block forms: column
    // component t of every element node).  Two contributions of one pair never share a slot, and lanes t != t' never
    // touch the same slot at all, so every slot still receives its terms in adjacency order.
    constexpr int L = CPP % NEN == 0 ? CPP / NEN : 1, CPL = CPP / L;
    if (tid < nrows * L && !(dbg & 1)) {
        const int row = tid / L, t = tid - row * L;
        const int pb = off[row], pe = off[row + 1];
        for (int i = pb; i < pe; ++i) {
            const int bv = i * CPP + row * PADV, bs = i * CPP + row * PADS;
            double vv[CPL];
            int ss[CPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                vv[j] = cval[bv + j * L + t];
                ss[j] = cslot[bs + j * L + t];
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[ss[j]] += vv[j];
        }
    }
    __syncthreads();
    for (int i = tid; i < ncols; i += 256) a.val[rs0 + i] = acc[i];
}

struct RhsArgs {
    const int32_t* conn;
    const int32_t* n2e_ptr;
    const int32_t* n2e;
    const double* xyz;
    double* rhs;
    int32_t n_own;
    int nen, dofs;
    double base[10];  // sum_q w_q phi_q,i
    double f[MAX_DOFS];
};

// load vector in two phases: |det B_e| of every element (coalesced over elements), then each owned
// node sums base[local index] * |det| over its adjacent elements in adjacency order
template <int DIM>
__global__ void k_elem_absdet(const int32_t* __restrict__ conn, int nen, const double* __restrict__ xyz,
                              int64_t n_elem, double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elem) return;
    double X[DIM + 1][DIM];
#pragma unroll
    for (int v = 0; v <= DIM; ++v) {
        const int32_t nd = conn[e * nen + v];
#pragma unroll
        for (int d = 0; d < DIM; ++d) X[v][d] = xyz[(int64_t)nd * DIM + d];
    }
    out[e] = fabs(affine_det<DIM>(X));
}

// RHS_NPB nodes per workgroup: the (node, element) pairs of the workgroup are one contiguous run
// of the adjacency, read coalesced with one lane per pair into an LDS park; then one lane per
// node adds its segment in adjacency order (the order does not depend on the launch shape).
constexpr int RHS_NPB = 64;

__global__ __launch_bounds__(256) void k_rhs(RhsArgs a, const double* __restrict__ absdet, int cap) {
    extern __shared__ double park[];
    const int32_t node0 = blockIdx.x * RHS_NPB;
    const int32_t node1 = min(a.n_own, node0 + RHS_NPB);
    const int32_t p0 = a.n2e_ptr[node0];
    const int tid = threadIdx.x;
    const int32_t node = node0 + tid;
    const bool mine = tid < RHS_NPB && node < node1;
    const int32_t nb = mine ? a.n2e_ptr[node] : 0, ne = mine ? a.n2e_ptr[node + 1] : 0;
    double sum = 0.0;
    // the run is processed in windows of `cap` pairs (one window unless a node has very many elements)
    for (int32_t w0 = p0; w0 < a.n2e_ptr[node1]; w0 += cap) {
        const int32_t w1 = min(a.n2e_ptr[node1], w0 + cap);
        // (four pairs per lane and trip: their adjacency entries, then their |det B|, requested together)
        for (int32_t q0 = w0 + tid; q0 < w1; q0 += 256 * 4) {
            int32_t idx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) idx[u] = a.n2e[min(q0 + 256 * u, w1 - 1)];
            double ad[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ad[u] = absdet[idx[u] / a.nen];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int32_t p = q0 + 256 * u;
                if (p < w1) {
                    const int li = idx[u] - (idx[u] / a.nen) * a.nen;
                    double b = 0.0;
                    for (int i = 0; i < 10; ++i) b = i == li ? a.base[i] : b;
                    park[p - w0] = b * ad[u];
                }
            }
        }
        __syncthreads();
        for (int32_t p = max(nb, w0); p < min(ne, w1); ++p) sum += park[p - w0];
        __syncthreads();
    }
    if (mine)
        for (int d = 0; d < a.dofs; ++d) a.rhs[(int64_t)node * a.dofs + d] = sum * a.f[d];
}

struct BcArgs {
    int n, dofs;
    int32_t flag[MAX_BC];
    int32_t mask[MAX_BC * MAX_DOFS];
    double value[MAX_BC * MAX_DOFS];
};

__global__ void k_dirichlet(BcArgs b, const int32_t* __restrict__ nflag, const int32_t* __restrict__ rowptr,
                            const int32_t* __restrict__ colind, double* val, double* rhs, int32_t* isdir,
                            int32_t n_rows) {
    const int32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const int32_t node = row / b.dofs;
    const int comp = row - node * b.dofs;
    const int32_t f = nflag[node];
    int hit = -1;
    for (int k = 0; k < b.n; ++k)
        if (hit < 0 && b.flag[k] == f && b.mask[k * b.dofs + comp]) hit = k;
    if (hit < 0) return;
    for (int32_t p = rowptr[row]; p < rowptr[row + 1]; ++p) val[p] = colind[p] == row ? 1.0 : 0.0;
    rhs[row] = b.value[hit * b.dofs + comp];
    isdir[row] = 1;
}

// per-node variant: node list + per-dof mask/value (what BCBuilder::setRHS obtains by evaluating
// the user's boundary function at every flagged unique node, BCBuilder_def.hpp:128-143)
__global__ void k_dirichlet_nodes(const int32_t* __restrict__ nodes, const int32_t* __restrict__ mask,
                                  const double* __restrict__ values, int32_t n, int dofs,
                                  const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, double* val,
                                  double* rhs, int32_t* isdir) {
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * dofs) return;
    const int32_t k = t / dofs;
    const int comp = t - k * dofs;
    if (mask && !mask[t]) return;
    const int32_t row = nodes[k] * dofs + comp;
    for (int32_t p = rowptr[row]; p < rowptr[row + 1]; ++p) val[p] = colind[p] == row ? 1.0 : 0.0;
    rhs[row] = values[t];
    isdir[row] = 1;
}

// ---------------------------------------------------------------------------------------------
// P2 elements, scalar forms (FE::assemblyLaplace / assemblyMass with the 10 x 10 -- 2D: 6 x 6 -- element matrices of
// FE_def.hpp:637-665, 485-499): element-major evaluation, ONE ELEMENT PER WAVEFRONT.  The quadrature weights and the reference
// basis values / gradients are staged in LDS once per workgroup; a wave loads its element's vertices, forms B^-1 and |det B|,
// parks the transformed gradients of all basis functions at all quadrature points in LDS (nq x NEN x DIM values), and its
// lanes then take the NEN^2 entries of the element matrix, which leave as one contiguous 800-byte stream.  The rows of the
// global matrix are then summed from these element matrices by the pair kernels (k_assemble_pairs reads row li of element e
// instead of re-deriving it: the pair kernels alone evaluate every P2 element ten times, 11.4 ms at a 64^3-cell cube against
// the figure in DESIGN.md section 4 with this kernel) -- in adjacency order, no atomics: bitwise reproducible as before.
// ---------------------------------------------------------------------------------------------
template <int DIM, int NEN, int FORM>
__global__ __launch_bounds__(256) void k_elem_matrix(AsmArgs a, int64_t n_elem, double* __restrict__ ke) {
    extern __shared__ double sm[];
    const int nq = a.nq, ntab = nq * (1 + NEN + NEN * DIM + DIM + 1);
    double* s_w = sm;
    double* s_phi = s_w + nq;
    double* s_dphi = s_phi + nq * NEN;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    double* Gs = sm + ntab + (ntab & 1) + (size_t)w * nq * NEN * DIM;       // this wave's transformed gradients [q][i][d]
    for (int i = tid; i < ntab; i += 256) sm[i] = a.tab[i];
    __syncthreads();
    const int64_t nwave = (int64_t)gridDim.x * 4;
    const int64_t trips = (n_elem + nwave - 1) / nwave;
    for (int64_t k = 0; k < trips; ++k) {
        const int64_t e = k * nwave + (int64_t)blockIdx.x * 4 + w;
        const bool on = e < n_elem;
        // vertices = the first DIM + 1 nodes of the element; lane l < (DIM + 1) DIM holds coordinate (l / DIM, l % DIM)
        double xv = 0.0;
        if (on && lane < (DIM + 1) * DIM) xv = a.xyz[(int64_t)a.conn[e * NEN + lane / DIM] * DIM + (lane % DIM)];
        double X[DIM + 1][DIM];
#pragma unroll
        for (int v = 0; v <= DIM; ++v)
#pragma unroll
            for (int d = 0; d < DIM; ++d) X[v][d] = __shfl(xv, v * DIM + d, 64);
        double absdet = 0.0;
        if constexpr (FORM == F_LAPLACE) {
            double Binv[DIM][DIM];
            absdet = on ? fabs(affine<DIM>(X, Binv)) : 0.0;
            if (on)
                for (int t = lane; t < nq * NEN; t += 64) {
                    double g[DIM];
                    grad_t<DIM, NEN>(s_dphi, t / NEN, t % NEN, Binv, g);
#pragma unroll
                    for (int d = 0; d < DIM; ++d) Gs[t * DIM + d] = g[d];
                }
        } else {
            absdet = on ? fabs(affine_det<DIM>(X)) : 0.0;
        }
        __syncthreads();
        if (on) {
            double* __restrict__ out = ke + e * (NEN * NEN);
            for (int t = lane; t < NEN * NEN; t += 64) {
                const int i = t / NEN, j = t - i * NEN;
                double v = 0.0;
                if constexpr (FORM == F_LAPLACE) {
                    for (int q = 0; q < nq; ++q) {
                        const double* gi = Gs + (q * NEN + i) * DIM;
                        const double* gj = Gs + (q * NEN + j) * DIM;
#pragma unroll
                        for (int d = 0; d < DIM; ++d) v += s_w[q] * gi[d] * gj[d];
                    }
                    out[t] = zero_small(a, v * absdet);
                } else {
                    for (int q = 0; q < nq; ++q) v += s_w[q] * s_phi[q * NEN + i] * s_phi[q * NEN + j];
                    out[t] = v * absdet - a.p0 * absdet * a.p1;
                }
            }
        }
        __syncthreads();        // Gs is rewritten by the next trip
    }
}

// element matrices of the whole mesh into c->d_ke (P2, F_LAPLACE / F_MASS); returns the args with ke set
template <int DIM, int NEN, int FORM>
int launch_elem_matrices(fedd_ctx* c, AsmArgs& a, int ntab) {
    FEDD_TRY(c->d_ke.ensure((size_t)c->n_elem * NEN * NEN));
    const size_t lds = ((size_t)ntab + (ntab & 1) + 4 * (size_t)a.nq * NEN * DIM) * sizeof(double);
    FEDD_CHECK(lds <= 64 * 1024, "element matrices: quadrature tables of %d points do not fit the LDS", a.nq);
    const int64_t nwg = std::min<int64_t>((c->n_elem + 3) / 4, 256 * 8);
    ScopedTimer t(c, FEDD_T_ASSEMBLE);
    hipLaunchKernelGGL((k_elem_matrix<DIM, NEN, FORM>), dim3((unsigned)nwg), dim3(256), lds, c->stream, a, c->n_elem, c->d_ke.p);
    t.stop();
    FEDD_HIP(hipGetLastError());
    a.ke = c->d_ke.p;
    return 0;
}

template <int DIM, int NEN, int FORM>
int launch_pairs(fedd_ctx* c, const AsmArgs& a, int ntab, int64_t n_rows, int rowcap) {
    constexpr int CPP = PairCfg<DIM, NEN, FORM>::CPP;
    const int maxdeg = std::max(1, c->max_deg);
    const size_t per_row = (size_t)maxdeg * CPP * 12;  // f64 value + i32 column per contribution
    // rows per workgroup from the LDS budget: 37 KB lets four workgroups share the 160 KB of a CU (40 KB: three;
    // P1 Laplace 3D: 32 rows, 0.81 -> 0.59 ms at 100^3 cells, 7.9 -> 5.7 ms at 214^3; smaller budgets gain nothing more)
    int R = (int)std::min<size_t>(63, ((size_t)c->asm_lds_kb * 1024) / per_row);  // <= 63: one wave scans the row offsets
    if (R < 1) R = 1;
    const int cap = R * maxdeg * CPP + R * 2;                   // + the per-row bank-shift padding
    const size_t lds = (size_t)ntab * 8 + (size_t)cap * 12 + (size_t)(R + 1) * 4 + 16;
    FEDD_CHECK(lds <= 160 * 1024, "assembly: a node with %d incident elements does not fit the LDS contribution buffer", maxdeg);
    int tl = 0;
    while ((1 << tl) < std::min(64, std::max(1, rowcap))) ++tl;
    auto kern = k_assemble_pairs<DIM, NEN, FORM>;
    if (lds > 64 * 1024)
        FEDD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((unsigned)((n_rows + R - 1) / R)), block(256);
    ScopedTimer t(c, FEDD_T_ASSEMBLE);
    hipLaunchKernelGGL(kern, grid, block, lds, c->stream, a, R, tl, cap);
    t.stop();
    FEDD_HIP(hipGetLastError());
    return 0;
}

// slot-addressed kernel; returns -1 (without error) when a row or the LDS budget does not suit it
template <int DIM, int NEN, int FORM>
int launch_slots(fedd_ctx* c, const AsmArgs& a, int ntab, int64_t n_rows, int rowcap) {
    constexpr int CPP = PairCfg<DIM, NEN, FORM>::CPP;
    const int maxdeg = std::max(1, c->max_deg);
    rowcap = std::max(1, rowcap);
    // per row: parked contributions (f64 value + u16 slot, + the bank shifts), image of the CSR row (f64 + i32)
    const size_t per_row = (size_t)maxdeg * CPP * 10 + 12 + (size_t)rowcap * 12 + 8;
    const size_t budget = (size_t)c->asm_lds_kb * 1024;
    // (R <= 63: one wave scans the row offsets, lane nrows writes the total)
    int R = (int)std::min<size_t>(63, budget > (size_t)ntab * 8 ? (budget - (size_t)ntab * 8) / per_row : 0);
    if (R < 1) R = (int)std::min<size_t>(63, (150 * 1024 - (size_t)ntab * 8) / per_row);   // large P2 rows: one workgroup per CU
    if (R < 1) return -1;
    const int cap_cols = R * rowcap;
    if (cap_cols > 65535) return -1;    // u16 positions
    int cap_contrib = R * maxdeg * CPP + R * 2 + 2;
    cap_contrib += cap_contrib & 1;     // keeps the arrays behind it 8-byte aligned
    const size_t lds = (size_t)ntab * 8 + (size_t)cap_contrib * 8 + (size_t)cap_cols * 12 + (size_t)(2 * R + 2) * 4 +
                       (size_t)cap_contrib * 2 + 16;
    if (lds > 160 * 1024) return -1;
    const int nwg = (int)((n_rows + R - 1) / R);
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            FEDD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ScopedTimer t(c, FEDD_T_ASSEMBLE);
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, c->stream, a, R, cap_contrib, cap_cols, nwg, c->asm_dbg);
        t.stop();
        return 0;
    };
    if (NEN <= 4 && c->asm_u >= 3) FEDD_TRY(go(k_assemble_slots<DIM, NEN, FORM, (NEN <= 4 ? 3 : 1)>));
    else if (NEN <= 4 && c->asm_u == 2) FEDD_TRY(go(k_assemble_slots<DIM, NEN, FORM, (NEN <= 4 ? 2 : 1)>));
    else FEDD_TRY(go(k_assemble_slots<DIM, NEN, FORM, 1>));
    FEDD_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Element-major tiles (asm_kind 4; P1 simplices, Laplace / vector Laplace / linear elasticity).
// FE::assemblyLaplace / assemblyLinElasXDim loop over ELEMENTS (FE_def.hpp:637-665, 2894-3031); the pair kernels above turn
// that inside out (one lane per (row, element) pair re-derives the element's geometry, ~4x per element, through a chain of
// three dependent gathers).  Here a workgroup owns a TILE: a compact cluster of ~27 nodes (a cell of a coordinate lattice) and
// every element that touches it.
//   phase 0: the coordinates of the tile's nodes and of their neighbours (the tile's "extended" node list) -> LDS
//            (independent loads: no chain); the tile's gather lists -> LDS;
//   phase 1: one lane per DISTINCT element of the tile: vertices by their extended-local ids (one 4-byte record), affine
//            map, transformed gradients; Laplace parks the 4 x 4 element matrix, elasticity the gradients and |det B|, in LDS.
//            Every element is evaluated once per tile it touches (~2.4x overall for 3^3-node tiles of the Kuhn cube);
//   phase 2: one lane per (row, CSR slot): adds the contributions of its GATHER LIST -- (element of the tile, local row,
//            local column) triples in the order of the node's sorted adjacency, i.e. the summation order of the pair
//            kernels -- and writes the slot.  No search, no sweep over the row's other contributions, no atomics:
//            bitwise reproducible.
// The tile structures (node lists, extended lists, element records, gather lists) depend on the mesh only, like the node ->
// element adjacency: they are built once per mesh, on the host from the adjacency the device built (build_tiles below,
// threads over tiles), at the first assembly that uses them.  Meshes whose tiles do not fit the limits (more than 255
// extended nodes or 448 elements after splitting), P2 elements and the other forms stay on the pair kernels.
// ---------------------------------------------------------------------------------------------
constexpr int TL_RMAX = 64;       // nodes of a tile
constexpr int TL_ELMAX = 448;     // distinct elements of a tile
constexpr int TL_EXTMAX = 255;    // extended nodes (8-bit local ids)

// Everything a tile needs lies in ONE contiguous blob of 32-bit words (a single streaming read per tile, no pointer chasing):
//   ext[NE]   extended node list: the tile's R nodes first (ascending), then the other vertices of its elements
//   nb[R]     node-level row start of every tile node (the pattern is a function of the mesh: symbolic.hip lays the dof rows
//             out in closed form from it, k_expand_pattern)
//   el[EL]    element records: NEN extended-local ids, one byte each
//   gp[R+1]   start of every node's gather entries within the tile's list
//   sp[R+1]   start of every node's slot offsets within the tile's slot array
//   gslot     16-bit: per node nslot + 1 offsets into its entries (padded to a word)
//   glist     16-bit: gather entries, element-of-tile << 4 | local row << 2 | local column (padded to a word)
struct TileHdr {
    uint32_t off;          // first word of the blob
    uint16_t R, NE, EL, NS;    // nodes, extended nodes, elements, 16-bit slot offsets
    uint16_t GN, MAXS;         // gather entries, most slots of a node
};

constexpr int TL_BLOBMAX = 4096;  // words of the largest blob this path takes (the next tile's blob waits in registers: BLOBMAX / BS per lane)

template <int DIM, int FORM, int BS, bool ZE = false /* doSetZeros thresholding compiled in (asm_zero_eps > 0) */>
__global__ __launch_bounds__(BS) void k_assemble_tiles(AsmArgs a, const TileHdr* __restrict__ hdr, const uint32_t* __restrict__ blob,
                                                        const uint32_t* __restrict__ shape_off,
                                                        int32_t ntile, int tiles_per_wg, int block_mode, int lds_el, int lds_blob, int dbg) {
    constexpr int NEN = DIM + 1, TL_PFW = (TL_BLOBMAX + BS - 1) / BS;
    constexpr int PARK = FORM == F_LAPLACE ? NEN * NEN : NEN * DIM + 1;     // element matrix | transformed gradients and |det B|
    extern __shared__ double sm[];
    const int ntab = a.nq * (1 + NEN + NEN * DIM + DIM + 1);
    double* s_w = sm;
    double* s_dphi = s_w + a.nq + a.nq * NEN;
    double* park = sm + ntab + (ntab & 1);                      // [lds_el][PARK]
    uint32_t* sb = reinterpret_cast<uint32_t*>(park + (size_t)lds_el * PARK);    // [lds_blob] the tile's blob (8-byte aligned)
    int32_t* pre = reinterpret_cast<int32_t*>(sb + lds_blob);   // [TL_RMAX + 1] prefix of the nodes' slot counts
    int32_t* heavy = pre + TL_RMAX + 2;                         // [TL_RMAX] slot with the longest gather list of every node
    const int tid = threadIdx.x;
    for (int i = tid; i < ntab; i += BS) sm[i] = a.tab[i];
    // park index of value c of element e: Laplace (16 values: a 128-byte element stride would put all lanes of a store on one
    // bank) value-major, elasticity (13 values, odd stride) element-major
    auto pix = [&](int e, int cidx) { return FORM == F_LAPLACE ? cidx * lds_el + e : e * PARK + cidx; };
    // A workgroup walks a contiguous run of tiles; the NEXT tile's blob is requested (into registers) before the current one is
    // worked on, its header one tile earlier still: nothing in the loop waits for a chain of dependent global loads.
    const int32_t t_begin = blockIdx.x * tiles_per_wg, t_end = min(ntile, t_begin + tiles_per_wg);
    if (t_begin >= t_end) return;
    TileHdr h = hdr[t_begin];
    TileHdr h_next = t_begin + 1 < t_end ? hdr[t_begin + 1] : h;
    auto blob_words = [&](const TileHdr& q) {
        return 2 * DIM * (int)q.NE + (int)q.NE + (int)q.R + (int)q.EL + 2 * ((int)q.R + 1) + (((int)q.NS + 1) >> 1) + (int)((q.GN + 1) >> 1);
    };
    // words of the blob that belong to this tile alone (coordinates and ids of its extended nodes, row starts); the rest -- element
    // records and gather lists in tile-local numbering, its SHAPE -- is read from the first tile of the same shape (shape_off,
    // build_tile_shapes) and stays in LDS while consecutive tiles share it
    auto own_words = [&](const TileHdr& q) { return 2 * DIM * (int)q.NE + (int)q.NE + (int)q.R; };
    uint32_t so = shape_off[t_begin];                       // first word of the current tile's shape
    uint32_t so_next = t_begin + 1 < t_end ? shape_off[t_begin + 1] : so;
    uint32_t so_lds = 0xffffffffu;                          // the shape whose words are in LDS
    uint32_t pf[TL_PFW];
    {
        const int nw = blob_words(h), w0 = own_words(h);
#pragma unroll
        for (int u = 0; u < TL_PFW; ++u) {
            const int i = tid + BS * u;
            pf[u] = i < nw ? blob[i < w0 ? (size_t)h.off + i : (size_t)so + (i - w0)] : 0u;
        }
    }
    const int dofs = a.dofs;
    const bool full = block_mode == FEDD_BLOCK_FULL;
    const int ncomp = dofs == 1 ? 1 : (full ? dofs * dofs : dofs);
    const double w0 = a.tab[0];
    for (int32_t tile = t_begin; tile < t_end; ++tile) {
        const int R = h.R, NE = h.NE, EL = h.EL;
        __syncthreads();        // the previous tile is done with sb / park / pre
        {
            const int keep_from = so == so_lds ? own_words(h) : lds_blob;     // (uniform) the shape words in LDS are this tile's
#pragma unroll
            for (int u = 0; u < TL_PFW; ++u)
                if (tid + BS * u < keep_from) sb[tid + BS * u] = pf[u];
            so_lds = so;
        }
        // request the next tile's blob (its shape only if it is another one), and the header of the tile after it
        const bool more = tile + 1 < t_end;
        const TileHdr hn = h_next;
        const uint32_t son = so_next;
        if (more) {
            const int w0 = own_words(hn), nw = son == so ? w0 : blob_words(hn);
#pragma unroll
            for (int u = 0; u < TL_PFW; ++u) {
                const int i = tid + BS * u;
                pf[u] = i < nw ? blob[i < w0 ? (size_t)hn.off + i : (size_t)son + (i - w0)] : 0u;
            }
            if (tile + 2 < t_end) {
                h_next = hdr[tile + 2];
                so_next = shape_off[tile + 2];
            }
        }
        const double* xs = reinterpret_cast<const double*>(sb);                  // [NE][DIM] coordinates of the extended nodes
        const int32_t* nbv = reinterpret_cast<const int32_t*>(sb) + 2 * DIM * NE + NE;
        const uint32_t* el = sb + 2 * DIM * NE + NE + R;
        const uint32_t* gp = el + EL;
        const uint32_t* sp = gp + R + 1;
        const uint16_t* gslot = reinterpret_cast<const uint16_t*>(sp + R + 1);
        const uint16_t* glist = gslot + 2 * ((h.NS + 1) >> 1);
        __syncthreads();
        // slots before node p: sp[p] counts nslot + 1 offsets per node, so pre[p] = sp[p] - p (no prefix sum in the kernel)
        if (tid <= R) pre[tid] = (int)sp[tid] - tid;
        // the slot of every node with the longest gather list (the diagonal: all 24 elements of a Kuhn-cube node against 4 - 6
        // of an edge): phase 2 hands these to its first lanes, so that one wave walks the long lists and the others the short
        // ones (a node's sixteen slots on sixteen consecutive lanes made EVERY wave wait for its diagonals: 3 rounds of 8 each)
        if (h.MAXS <= 16)       // (sixteen lanes per node: one list length each, the longest by four exchanges; ties: lowest slot)
            for (int q = tid; q < ((R * 16 + 63) & ~63); q += BS) {
                const int p = min(q >> 4, R - 1), sl = q & 15;
                const int s0 = (int)sp[p], ns = (int)sp[p + 1] - s0 - 1;
                int key = sl < ns ? (((int)gslot[s0 + sl + 1] - (int)gslot[s0 + sl]) << 4) + (15 - sl) : -1;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) key = max(key, __shfl_xor(key, off, 64));
                if (sl == 0 && (q >> 4) < R) heavy[p] = key < 0 ? 0 : 15 - (key & 15);
            }
        // ---- phase 1: the elements of the tile, once each ----
        for (int e = tid; e < ((dbg & 1) ? 0 : EL); e += BS) {
            const uint32_t rec = el[e];
            double X[NEN][DIM];
#pragma unroll
            for (int v = 0; v < NEN; ++v) {
                const int li = (rec >> (8 * v)) & 255;
#pragma unroll
                for (int d = 0; d < DIM; ++d) X[v][d] = xs[li * DIM + d];
            }
            double Binv[DIM][DIM];
            const double absdet = fabs(affine<DIM>(X, Binv));
            double G[NEN][DIM];
#pragma unroll
            for (int j = 0; j < NEN; ++j) grad_t<DIM, NEN>(s_dphi, 0, j, Binv, G[j]);
            // parked value-major / element-minor: consecutive lanes (elements) write consecutive words (an element-major park
            // with its 128-byte stride put all 64 lanes of a store on one bank: phase 1 took 3.7 instead of 0.7 ms at cfg 3)
            double* pk = park;
            if constexpr (FORM == F_LAPLACE) {
                // row i of the element matrix as compute_pair forms it (FE_def.hpp:637-656): wg = w g_i, v_j = sum_d wg_d g_jd
#pragma unroll
                for (int i = 0; i < NEN; ++i) {
                    double wg[DIM];
#pragma unroll
                    for (int d = 0; d < DIM; ++d) wg[d] = w0 * G[i][d];
#pragma unroll
                    for (int j = 0; j < NEN; ++j) {
                        double v = 0.0;
#pragma unroll
                        for (int d = 0; d < DIM; ++d) v += wg[d] * G[j][d];
                        pk[pix(e, i * NEN + j)] = ZE ? zero_small(a, v * absdet) : v * absdet;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NEN; ++j)
#pragma unroll
                    for (int d = 0; d < DIM; ++d) pk[pix(e, j * DIM + d)] = G[j][d];
                pk[pix(e, NEN * DIM)] = absdet;
            }
        }
        __syncthreads();
        // ---- phase 2: one lane per (node, slot[, row component, column component]) ----
        // item -> (node, slot): 16 slot places per node where no node of the tile has more (the structured grids: 15), else
        // through the prefix of the slot counts
        const bool direct = h.MAXS <= 16;
        const int nitem = (direct ? R * 16 : pre[R]) * ncomp;
        for (int item = tid; item < nitem; item += BS) {
            const int q = item / ncomp, ab = item - q * ncomp;
            int p, sl;
            if (direct) {
                if (q < R) {            // the nodes' longest lists first
                    p = q;
                    sl = heavy[p];
                    if (pre[p + 1] - pre[p] <= 0) continue;
                } else {                // then the other slots, fifteen places per node
                    const int q2 = q - R;
                    p = q2 / 15;
                    const int s2 = q2 - p * 15, hv = heavy[p];
                    sl = s2 + (s2 >= hv ? 1 : 0);
                    if (sl >= pre[p + 1] - pre[p]) continue;
                }
            } else {
                int lo = 0, hi = R - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (pre[mid] <= q) lo = mid;
                    else hi = mid - 1;
                }
                p = lo;
                sl = q - pre[p];
            }
            const int nslot = pre[p + 1] - pre[p];
            const int ca = ncomp == 1 ? 0 : (full ? ab / dofs : ab);        // row component
            const int cb = ncomp == 1 ? 0 : (full ? ab - ca * dofs : ab);   // column component
            const uint32_t b = gp[p] + gslot[sp[p] + sl], eend = (dbg & 2) ? b : gp[p] + gslot[sp[p] + sl + 1];
            double acc = 0.0;
            // eight gather entries at a time: their ids, then their values, as independent LDS reads; added in list order
            for (uint32_t k = b; k < eend; k += 8) {
                uint32_t en[8];
                double val[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) en[u] = k + u < eend ? glist[k + u] : 0xffffffffu;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t e8 = en[u] == 0xffffffffu ? 0u : en[u];
                    const int eloc = e8 >> 4, li = (e8 >> 2) & 3, j = e8 & 3;
                    const double* pk = park;
                    if constexpr (FORM == F_LAPLACE) {
                        val[u] = pk[pix(eloc, li * NEN + j)];
                    } else {
                        // 2 mu eps_i:eps_j + lam tr(eps_i) tr(eps_j), compute_pair's expression (FE_def.hpp:2894-3031; :4931-4944)
                        const double lam = a.p0, mu = a.p1;
                        double dot = 0.0;
#pragma unroll
                        for (int d = 0; d < DIM; ++d) dot += pk[pix(eloc, li * DIM + d)] * pk[pix(eloc, j * DIM + d)];
                        const double gia = pk[pix(eloc, li * DIM + ca)], gja = pk[pix(eloc, j * DIM + ca)];
                        const double gib = pk[pix(eloc, li * DIM + cb)], gjb = pk[pix(eloc, j * DIM + cb)];
                        const double vb = w0 * (mu * ((cb == ca ? dot : 0.0) + gib * gja) + lam * gia * gjb);
                        val[u] = pk[pix(eloc, NEN * DIM)] * vb;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += en[u] == 0xffffffffu ? 0.0 : val[u];
            }
            // the row's place in the CSR arrays, closed form from the node-level row start (symbolic.hip k_expand_pattern)
            const int32_t nb = nbv[p];
            const int32_t start = dofs == 1 ? nb : (full ? nb * dofs * dofs + ca * nslot * dofs : nb * dofs + ca * nslot);
            a.val[start + (full ? sl * dofs + cb : sl)] = acc;
        }
        h = hn;
        so = son;
    }
}

// ---- shapes: tiles whose element records and gather lists agree word for word (all interior tiles of a structured grid) share them ----
// shape_off[t] = first word of tile t's shape part: its own (hdr[t].off + own words), or that of the first tile with the same
// header counts and the same shape words.  k_tile_shape_hash: one wave per tile hashes counts and words, the tile with the
// smallest id claims the hash; k_tile_shape_pick: every tile compares itself word by word with the claimant (a hash collision
// or a tile of another shape keeps its own).  The kernel above then streams 3.6 instead of 10.5 KB per tile of the Kuhn cube.
__device__ __forceinline__ uint64_t tl_mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
__device__ __forceinline__ int tl_own_words(const TileHdr& q, int dim) { return 2 * dim * (int)q.NE + (int)q.NE + (int)q.R; }
__device__ __forceinline__ int tl_all_words(const TileHdr& q, int dim) {
    return tl_own_words(q, dim) + (int)q.EL + 2 * ((int)q.R + 1) + (((int)q.NS + 1) >> 1) + (int)((q.GN + 1) >> 1);
}
__global__ __launch_bounds__(256) void k_tile_shape_hash(const TileHdr* __restrict__ hdr, const uint32_t* __restrict__ blob, int32_t ntile, int dim,
                                                         unsigned long long* __restrict__ tkey, int32_t* __restrict__ trep, uint32_t tmask,
                                                         unsigned long long* __restrict__ hash) {
    const int32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= ntile) return;
    const TileHdr h = hdr[t];
    const int w0 = tl_own_words(h, dim), w1 = tl_all_words(h, dim);
    uint64_t acc = 0;
    for (int i = w0 + lane; i < w1; i += 64) acc += tl_mix(((uint64_t)(i - w0 + 1) << 32) ^ blob[(size_t)h.off + i]);      // order-free sum of position-keyed words
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down((unsigned long long)acc, off, 64);
    if (lane != 0) return;
    uint64_t key = tl_mix(acc ^ ((uint64_t)h.R << 48) ^ ((uint64_t)h.NE << 32) ^ ((uint64_t)h.EL << 16) ^ (uint64_t)h.NS ^ ((uint64_t)h.GN << 24) ^ ((uint64_t)h.MAXS << 56));
    if (key == 0) key = 1;
    hash[t] = key;
    uint32_t slot = (uint32_t)(key >> 20) & tmask;
    for (uint32_t probe = 0; probe <= tmask; ++probe) {
        const unsigned long long prev = atomicCAS(&tkey[slot], 0ull, (unsigned long long)key);
        if (prev == 0ull || prev == key) {
            atomicMin(&trep[slot], t);
            return;
        }
        slot = (slot + 1) & tmask;
    }
}
__global__ __launch_bounds__(256) void k_tile_shape_pick(const TileHdr* __restrict__ hdr, const uint32_t* __restrict__ blob, int32_t ntile, int dim,
                                                         const unsigned long long* __restrict__ tkey, const int32_t* __restrict__ trep, uint32_t tmask,
                                                         const unsigned long long* __restrict__ hash, uint32_t* __restrict__ shape_off,
                                                         int32_t* __restrict__ n_shared) {
    const int32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= ntile) return;
    const TileHdr h = hdr[t];
    const int w0 = tl_own_words(h, dim), w1 = tl_all_words(h, dim);
    const unsigned long long key = hash[t];
    uint32_t slot = (uint32_t)(key >> 20) & tmask;
    int32_t r = t;
    for (uint32_t probe = 0; probe <= tmask; ++probe) {
        const unsigned long long k2 = tkey[slot];
        if (k2 == key) {
            r = trep[slot];
            break;
        }
        if (k2 == 0ull) break;
        slot = (slot + 1) & tmask;
    }
    bool same = r != t;
    TileHdr hr = h;
    if (same) {
        hr = hdr[r];
        same = hr.R == h.R && hr.NE == h.NE && hr.EL == h.EL && hr.NS == h.NS && hr.GN == h.GN && hr.MAXS == h.MAXS;
    }
    if (same) {
        uint32_t diff = 0;
        for (int i = w0 + lane; i < w1; i += 64) diff |= blob[(size_t)h.off + i] ^ blob[(size_t)hr.off + i];
        same = __ballot(diff != 0) == 0ull;
    }
    if (lane == 0) {
        shape_off[t] = (same ? hr.off : h.off) + (uint32_t)w0;
        if (same) atomicAdd(n_shared, 1);
    }
}

static int build_tile_shapes(fedd_ctx* c) {
    const int32_t nt = (int32_t)c->tl_ntile;
    FEDD_TRY(c->tl_shape.ensure((size_t)nt));
    uint32_t tsize = 1024;
    while (tsize < 2u * (uint32_t)nt) tsize <<= 1;
    // scratch: hashes [nt] | table keys [tsize] (64-bit), claimants [tsize] + counter (32-bit)
    FEDD_TRY(c->d_cs_hash.ensure((size_t)nt + tsize));
    FEDD_TRY(c->d_itmp0.ensure((size_t)tsize + 1));
    unsigned long long* hash = (unsigned long long*)c->d_cs_hash.p;
    unsigned long long* tkey = hash + nt;
    int32_t* trep = c->d_itmp0.p;
    FEDD_HIP(hipMemsetAsync(tkey, 0, (size_t)tsize * sizeof(unsigned long long), c->stream));
    FEDD_HIP(hipMemsetAsync(trep, 0x7f, (size_t)tsize * sizeof(int32_t), c->stream));
    FEDD_HIP(hipMemsetAsync(trep + tsize, 0, sizeof(int32_t), c->stream));
    const dim3 g((unsigned)((nt + 3) / 4)), b(256);
    hipLaunchKernelGGL(k_tile_shape_hash, g, b, 0, c->stream, (const TileHdr*)c->tl_hdr.p, (const uint32_t*)c->tl_blob.p, nt, c->dim, tkey, trep, tsize - 1, hash);
    hipLaunchKernelGGL(k_tile_shape_pick, g, b, 0, c->stream, (const TileHdr*)c->tl_hdr.p, (const uint32_t*)c->tl_blob.p, nt, c->dim,
                       (const unsigned long long*)tkey, (const int32_t*)trep, tsize - 1, (const unsigned long long*)hash, c->tl_shape.p, trep + tsize);
    int32_t ns = 0;
    FEDD_HIP(hipMemcpyAsync(&ns, trep + tsize, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    c->tl_nshared = ns;
    FEDD_HIP(hipGetLastError());
    return 0;
}

// the tile structures of the current mesh (host; threads over tiles).  c->tl_state = -1 when the mesh does not fit.
static int build_tiles(fedd_ctx* c) {
    const int dim = c->dim, nen = c->nen;
    const int64_t nn = c->n_own + c->n_rowg;          // nodes with rows
    c->tl_state = -1;
    if (nen != dim + 1 || nn <= 0 || c->n_elem <= 0) return 0;
    std::vector<int32_t> conn((size_t)c->n_elem * nen), n2e_ptr((size_t)nn + 1);
    std::vector<double> xyz((size_t)c->n_node * dim);
    FEDD_HIP(hipMemcpyAsync(conn.data(), c->d_conn.p, conn.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipMemcpyAsync(n2e_ptr.data(), c->d_n2e_ptr.p, n2e_ptr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipMemcpyAsync(xyz.data(), c->d_xyz.p, xyz.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    std::vector<int32_t> n2e((size_t)n2e_ptr[(size_t)nn]);
    FEDD_HIP(hipMemcpyAsync(n2e.data(), c->d_n2e.p, n2e.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    // ---- nodes -> cells of a coordinate lattice with ~27 nodes each ----
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < nn; ++i)
        for (int d = 0; d < dim; ++d) {
            lo[d] = std::min(lo[d], xyz[(size_t)i * dim + d]);
            hi[d] = std::max(hi[d], xyz[(size_t)i * dim + d]);
        }
    double V = 1.0;
    for (int d = 0; d < dim; ++d) V *= std::max(hi[d] - lo[d], 1e-300);
    const double target = dim == 3 ? 27.0 : 25.0;
    const double w = std::pow(V * target / (double)nn, 1.0 / dim);
    int g[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d) g[d] = std::max(1, (int)std::floor((hi[d] - lo[d]) / w + 0.5));
    const int64_t ncell = (int64_t)g[0] * g[1] * g[2];
    if (ncell > ((int64_t)1 << 31) - 2) return 0;
    std::vector<int32_t> cell((size_t)nn), cnt((size_t)ncell + 1, 0);
    for (int64_t i = 0; i < nn; ++i) {
        int64_t id = 0, mul = 1;
        for (int d = 0; d < dim; ++d) {
            const double L = hi[d] - lo[d];
            int k = L > 0 ? (int)std::floor((xyz[(size_t)i * dim + d] - lo[d]) / L * g[d]) : 0;
            k = std::min(g[d] - 1, std::max(0, k));
            id += mul * k;
            mul *= g[d];
        }
        cell[(size_t)i] = (int32_t)id;
        ++cnt[(size_t)id + 1];
    }
    for (int64_t k = 0; k < ncell; ++k) cnt[(size_t)k + 1] += cnt[(size_t)k];
    std::vector<int32_t> order((size_t)nn);
    {
        std::vector<int32_t> pos(cnt.begin(), cnt.end() - 1);
        for (int64_t i = 0; i < nn; ++i) order[(size_t)pos[(size_t)cell[(size_t)i]]++] = (int32_t)i;   // ascending within a cell
    }
    // node-level row starts: the pattern's row of a node holds its distinct neighbours (itself included)
    std::vector<int32_t> nslot_of((size_t)nn), nb_of((size_t)nn + 1, 0);
    const unsigned nthr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto parallel = [&](int64_t n, auto&& body) {
        std::vector<std::thread> th;
        const int64_t chunk = (n + nthr - 1) / nthr;
        for (unsigned q = 0; q < nthr; ++q) {
            const int64_t t0 = (int64_t)q * chunk, t1 = std::min(n, t0 + chunk);
            if (t0 < t1) th.emplace_back(body, t0, t1);
        }
        for (auto& x : th) x.join();
    };
    parallel(nn, [&](int64_t i0, int64_t i1) {
        std::vector<int32_t> nbr;
        for (int64_t i = i0; i < i1; ++i) {
            nbr.clear();
            for (int32_t q = n2e_ptr[(size_t)i]; q < n2e_ptr[(size_t)i + 1]; ++q) {
                const int32_t el = n2e[(size_t)q] / nen;
                for (int j = 0; j < nen; ++j) nbr.push_back(conn[(size_t)el * nen + j]);
            }
            std::sort(nbr.begin(), nbr.end());
            nslot_of[(size_t)i] = (int32_t)(std::unique(nbr.begin(), nbr.end()) - nbr.begin());
        }
    });
    for (int64_t i = 0; i < nn; ++i) nb_of[(size_t)i + 1] = nb_of[(size_t)i] + nslot_of[(size_t)i];
    // ---- tiles: non-empty cells, cut into pieces that respect the limits ----
    struct Piece { int32_t b, e; };
    std::vector<Piece> pieces;
    auto distinct = [&](int32_t b, int32_t e, std::vector<int32_t>& els, std::vector<int32_t>& ext) {
        els.clear();
        for (int32_t k = b; k < e; ++k) {
            const int32_t nd = order[(size_t)k];
            for (int32_t p = n2e_ptr[(size_t)nd]; p < n2e_ptr[(size_t)nd + 1]; ++p) els.push_back(n2e[(size_t)p] / nen);
        }
        std::sort(els.begin(), els.end());
        els.erase(std::unique(els.begin(), els.end()), els.end());
        ext.clear();
        for (int32_t el : els)
            for (int j = 0; j < nen; ++j) ext.push_back(conn[(size_t)el * nen + j]);
        std::sort(ext.begin(), ext.end());
        ext.erase(std::unique(ext.begin(), ext.end()), ext.end());
    };
    {
        // (cells in parallel; the pieces of a cell stay together and in order)
        std::vector<std::vector<Piece>> per_thread(nthr);
        std::atomic<int> bad{0};
        std::vector<int64_t> bounds(nthr + 1, 0);
        for (unsigned q = 0; q <= nthr; ++q) bounds[q] = std::min<int64_t>(ncell, (int64_t)q * ((ncell + nthr - 1) / nthr));
        std::vector<std::thread> th;
        for (unsigned q = 0; q < nthr; ++q)
            th.emplace_back([&, q]() {
                std::vector<int32_t> els, ext;
                std::vector<Piece> stack;
                for (int64_t k = bounds[q]; k < bounds[q + 1]; ++k) {
                    if (cnt[(size_t)k + 1] == cnt[(size_t)k]) continue;
                    stack.push_back({cnt[(size_t)k], cnt[(size_t)k + 1]});
                    while (!stack.empty()) {
                        const Piece pc = stack.back();
                        stack.pop_back();
                        bool ok = pc.e - pc.b <= TL_RMAX;
                        if (ok) {
                            distinct(pc.b, pc.e, els, ext);
                            ok = (int)els.size() <= TL_ELMAX && (int)ext.size() <= TL_EXTMAX;
                        }
                        if (ok) per_thread[q].push_back(pc);
                        else if (pc.e - pc.b == 1) { bad = 1; }      // a single node that does not fit: pair kernels
                        else {
                            const int32_t mid = pc.b + (pc.e - pc.b) / 2;
                            stack.push_back({mid, pc.e});
                            stack.push_back({pc.b, mid});
                        }
                    }
                }
            });
        for (auto& x : th) x.join();
        if (bad) return 0;
        for (auto& v : per_thread) pieces.insert(pieces.end(), v.begin(), v.end());
    }
    const int64_t ntile = (int64_t)pieces.size();
    // ---- per tile: sizes, then the blobs (two parallel passes over the tiles) ----
    std::vector<TileHdr> hdr((size_t)ntile);
    std::vector<uint64_t> woff((size_t)ntile + 1, 0);
    std::vector<uint32_t> blob;
    std::atomic<int> failed{0};
    auto run_pass = [&](bool fill) {
        parallel(ntile, [&](int64_t t0, int64_t t1) {
            std::vector<int32_t> els, ext, extl, nbr;
            std::vector<uint16_t> ent, slots;
            std::vector<uint32_t> scount, gpv, spv;
            for (int64_t tI = t0; tI < t1; ++tI) {
                const Piece pc = pieces[(size_t)tI];
                const int R = pc.e - pc.b;
                distinct(pc.b, pc.e, els, ext);
                extl.assign(order.begin() + pc.b, order.begin() + pc.e);
                for (int32_t nd : ext)
                    if (!std::binary_search(order.begin() + pc.b, order.begin() + pc.e, nd)) extl.push_back(nd);
                auto ext_local = [&](int32_t nd) -> uint32_t {
                    auto it = std::lower_bound(order.begin() + pc.b, order.begin() + pc.e, nd);
                    if (it != order.begin() + pc.e && *it == nd) return (uint32_t)(it - (order.begin() + pc.b));
                    return (uint32_t)(std::lower_bound(extl.begin() + R, extl.end(), nd) - extl.begin());
                };
                const int NE = (int)extl.size(), EL = (int)els.size();
                uint32_t NS = 0, GN = 0, MAXS = 0;
                for (int p = 0; p < R; ++p) {
                    const int32_t nd = order[(size_t)pc.b + p];
                    MAXS = std::max<uint32_t>(MAXS, (uint32_t)nslot_of[(size_t)nd]);
                    NS += (uint32_t)nslot_of[(size_t)nd] + 1;
                    GN += (uint32_t)(n2e_ptr[(size_t)nd + 1] - n2e_ptr[(size_t)nd]) * nen;
                }
                if (NS > 65535 || GN > 65535 || EL > 4095) failed = 1;     // 16-bit offsets, 12-bit element-of-tile ids
                const uint64_t nw_raw = (uint64_t)2 * dim * NE + NE + R + EL + 2 * (uint64_t)(R + 1) + ((NS + 1) >> 1) + ((GN + 1) >> 1);
                const uint64_t nw = (nw_raw + 1) & ~(uint64_t)1;     // blobs start 8-byte aligned (the coordinates lead)
                if (nw_raw > TL_BLOBMAX) failed = 1;
                if (!fill) {
                    hdr[(size_t)tI] = TileHdr{0, (uint16_t)R, (uint16_t)NE, (uint16_t)EL, (uint16_t)NS, (uint16_t)GN, (uint16_t)MAXS};
                    woff[(size_t)tI + 1] = nw;
                    continue;
                }
                if (failed) return;
                uint32_t* wb0 = blob.data() + woff[(size_t)tI];
                double* cw = reinterpret_cast<double*>(wb0);
                for (int i = 0; i < NE; ++i)
                    for (int d = 0; d < dim; ++d) cw[(size_t)i * dim + d] = xyz[(size_t)extl[(size_t)i] * dim + d];
                uint32_t* wb = wb0 + 2 * dim * NE;
                for (int i = 0; i < NE; ++i) wb[i] = (uint32_t)extl[(size_t)i];
                for (int p = 0; p < R; ++p) wb[NE + p] = (uint32_t)nb_of[(size_t)order[(size_t)pc.b + p]];
                uint32_t* elw = wb + NE + R;
                for (int q = 0; q < EL; ++q) {
                    uint32_t rec = 0;
                    for (int j = 0; j < nen; ++j) rec |= ext_local(conn[(size_t)els[(size_t)q] * nen + j]) << (8 * j);
                    elw[q] = rec;
                }
                uint32_t* gpw = elw + EL;
                uint32_t* spw = gpw + R + 1;
                uint16_t* gsw = reinterpret_cast<uint16_t*>(spw + R + 1);
                uint16_t* glw = gsw + 2 * ((NS + 1) >> 1);
                uint32_t gpos = 0, spos = 0;
                for (int p = 0; p < R; ++p) {
                    const int32_t nd = order[(size_t)pc.b + p];
                    const int32_t pb = n2e_ptr[(size_t)nd], pe = n2e_ptr[(size_t)nd + 1];
                    nbr.clear();
                    for (int32_t q = pb; q < pe; ++q) {
                        const int32_t el = n2e[(size_t)q] / nen;
                        for (int j = 0; j < nen; ++j) nbr.push_back(conn[(size_t)el * nen + j]);
                    }
                    std::sort(nbr.begin(), nbr.end());
                    nbr.erase(std::unique(nbr.begin(), nbr.end()), nbr.end());
                    const int nslot = (int)nbr.size();
                    scount.assign((size_t)nslot + 1, 0);
                    for (int32_t q = pb; q < pe; ++q) {
                        const int32_t el = n2e[(size_t)q] / nen;
                        for (int j = 0; j < nen; ++j)
                            ++scount[(size_t)(std::lower_bound(nbr.begin(), nbr.end(), conn[(size_t)el * nen + j]) - nbr.begin()) + 1];
                    }
                    for (int sI = 0; sI < nslot; ++sI) scount[(size_t)sI + 1] += scount[(size_t)sI];
                    gpw[p] = gpos;
                    spw[p] = spos;
                    for (int sI = 0; sI <= nslot; ++sI) gsw[spos + sI] = (uint16_t)scount[(size_t)sI];
                    for (int32_t q = pb; q < pe; ++q) {       // adjacency order, then local column order: the pair kernels' order
                        const int32_t idx = n2e[(size_t)q], el = idx / nen, li = idx - el * nen;
                        const uint32_t eloc = (uint32_t)(std::lower_bound(els.begin(), els.end(), el) - els.begin());
                        for (int j = 0; j < nen; ++j) {
                            const size_t sI = (size_t)(std::lower_bound(nbr.begin(), nbr.end(), conn[(size_t)el * nen + j]) - nbr.begin());
                            glw[gpos + scount[sI]++] = (uint16_t)((eloc << 4) | ((uint32_t)li << 2) | (uint32_t)j);
                        }
                    }
                    gpos += (uint32_t)(pe - pb) * nen;
                    spos += (uint32_t)nslot + 1;
                }
                gpw[R] = gpos;
                spw[R] = spos;
            }
        });
    };
    run_pass(false);
    if (failed) return 0;
    int max_el = 0, max_ext = 0;
    uint64_t max_blob = 0;
    for (int64_t tI = 0; tI < ntile; ++tI) {
        max_el = std::max<int>(max_el, hdr[(size_t)tI].EL);
        max_ext = std::max<int>(max_ext, hdr[(size_t)tI].NE);
        max_blob = std::max(max_blob, woff[(size_t)tI + 1]);
        woff[(size_t)tI + 1] += woff[(size_t)tI];
    }
    if (woff[(size_t)ntile] >= ((uint64_t)1 << 32)) return 0;      // 32-bit word offsets
    for (int64_t tI = 0; tI < ntile; ++tI) hdr[(size_t)tI].off = (uint32_t)woff[(size_t)tI];
    blob.assign((size_t)woff[(size_t)ntile] + 1, 0u);
    run_pass(true);
    FEDD_TRY(c->tl_hdr.ensure(std::max<size_t>(1, (size_t)ntile * sizeof(TileHdr) / sizeof(uint32_t))));
    FEDD_TRY(c->tl_blob.ensure(blob.size()));
    FEDD_HIP(hipMemcpyAsync(c->tl_hdr.p, hdr.data(), (size_t)ntile * sizeof(TileHdr), hipMemcpyHostToDevice, c->stream));
    FEDD_HIP(hipMemcpyAsync(c->tl_blob.p, blob.data(), blob.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    c->tl_ntile = ntile;
    c->tl_max_el = max_el;
    c->tl_max_ext = max_ext;
    c->tl_max_blob = (int)max_blob;
    c->tl_state = 1;
    return 0;
}


// ---------------------------------------------------------------------------------------------
// The same tile structures built ON THE DEVICE (default since round 4; option "asm_tiles_host" 1 = the host builder above):
// no copy of the mesh back to the host, no host pass.  Cell binning by counting sort (k_tb_cell / k_tb_fill / k_tb_sort_cells),
// then one workgroup per tile (k_tb_build): the distinct elements and vertices of the tile by two bitonic sorts in LDS, the
// per-slot gather lists by a wave per node (a lane per CSR slot walks the node's adjacency once to count and once to fill: the
// entries of a slot stay in adjacency order, the summation order of the pair kernels), the blob put together in LDS and written
// as one contiguous stream.  A sizes pass (FILL = false) runs first; tiles that do not fit the limits are split in halves
// (k_tb_split_*) and sized again, as the host builder's recursion does.  The slot of a neighbour is its position in the
// node's row of the CURRENT pattern (columns sorted): the node-level pattern is a function of the mesh alone.
// ---------------------------------------------------------------------------------------------
constexpr int TB_BS = 256;        // threads of the build kernel
constexpr int TB_SORT = 2048;     // capacity of the LDS sorts: adjacency entries / element vertices of a tile
constexpr int TB_NSMAX = 64;      // most CSR slots of a node (a wave takes a node, a lane a slot)

struct TbGeom {
    double lo[3], L[3];
    int g[3];
};

template <int DIM>
__global__ void k_tb_cell(const double* __restrict__ xyz, int32_t nn, TbGeom gm, int32_t* __restrict__ cell, int32_t* __restrict__ cnt) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    int64_t id = 0, mul = 1;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        int k = gm.L[d] > 0 ? (int)floor((xyz[(size_t)i * DIM + d] - gm.lo[d]) / gm.L[d] * gm.g[d]) : 0;
        k = min(gm.g[d] - 1, max(0, k));
        id += mul * k;
        mul *= gm.g[d];
    }
    cell[i] = (int32_t)id;
    atomicAdd(&cnt[id], 1);
}

__global__ void k_tb_fill(const int32_t* __restrict__ cell, int32_t nn, int32_t* __restrict__ cursor, int32_t* __restrict__ order) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nn) order[atomicAdd(&cursor[cell[i]], 1)] = i;
}

// a wave per cell: its nodes ascending (rank sort through LDS; longer cells in global memory, lane 0)
__global__ __launch_bounds__(64) void k_tb_sort_cells(const int32_t* __restrict__ ptr, int32_t ncell, int32_t* order) {
    __shared__ int32_t sh[1024];
    const int32_t b = ptr[blockIdx.x], e = ptr[blockIdx.x + 1];
    const int n = e - b, lane = threadIdx.x;
    if (n <= 1) return;
    if (n > 1024) {
        if (lane == 0)
            for (int32_t i = b + 1; i < e; ++i) {
                const int32_t v = order[i];
                int32_t j = i - 1;
                while (j >= b && order[j] > v) {
                    order[j + 1] = order[j];
                    --j;
                }
                order[j + 1] = v;
            }
        return;
    }
    for (int i = lane; i < n; i += 64) sh[i] = order[b + i];
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        const int32_t v = sh[i];
        int rank = 0;
        for (int k = 0; k < n; ++k) rank += sh[k] < v ? 1 : 0;
        order[b + rank] = v;
    }
}

__global__ void k_tb_flag_cells(const int32_t* __restrict__ ptr, int32_t ncell, int32_t* __restrict__ flag) {
    const int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < ncell) flag[k] = ptr[k + 1] > ptr[k] ? 1 : 0;
}

__global__ void k_tb_pieces0(const int32_t* __restrict__ ptr, int32_t ncell, const int32_t* __restrict__ pos, int2* __restrict__ pieces) {
    const int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < ncell && ptr[k + 1] > ptr[k]) pieces[pos[k]] = make_int2(ptr[k], ptr[k + 1]);
}

__global__ void k_tb_split_count(const int32_t* __restrict__ split, int32_t n, int32_t* __restrict__ cnt) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cnt[i] = split[i] ? 2 : 1;
}

__global__ void k_tb_split_scatter(const int2* __restrict__ in, const int32_t* __restrict__ split, const int32_t* __restrict__ pos,
                                   int32_t n, int2* __restrict__ out) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 pc = in[i];
    if (split[i]) {
        const int32_t mid = pc.x + (pc.y - pc.x) / 2;
        out[pos[i]] = make_int2(pc.x, mid);
        out[pos[i] + 1] = make_int2(mid, pc.y);
    } else {
        out[pos[i]] = pc;
    }
}

__global__ void k_tb_set_off(TileHdr* __restrict__ hdr, const int64_t* __restrict__ off, int32_t n) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hdr[i].off = (uint32_t)off[i];
}

// ascending bitonic sort of s[0, N), N a power of two, by the whole workgroup
__device__ __forceinline__ void tb_bitonic(int32_t* s, int N, int tid) {
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (N >> 1); i += TB_BS) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo + j;
                const bool up = (lo & k) == 0;
                const int32_t a = s[lo], b = s[hi];
                if ((a > b) == up) {
                    s[lo] = b;
                    s[hi] = a;
                }
            }
            __syncthreads();
        }
}

// exclusive prefix of one value per thread over the workgroup (sh: TB_BS + 1 ints); *total = the sum
__device__ __forceinline__ int tb_scan(int v, int32_t* sh, int tid, int* total) {
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < TB_BS; off <<= 1) {
        const int t = tid >= off ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    const int incl = sh[tid];
    *total = sh[TB_BS - 1];
    __syncthreads();
    return incl - v;
}

__device__ __forceinline__ int tb_find(const int32_t* s, int n, int32_t v) {   // position of v in the ascending s[0, n), -1 if absent
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo < n && s[lo] == v ? lo : -1;
}

__device__ __forceinline__ int tb_upper(const int32_t* pre, int n, int t) {    // the p in [0, n) with pre[p] <= t < pre[p + 1]
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pre[mid] <= t) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// counters: 0 = pieces to split, 1 = a single node that does not fit (the mesh stays on the pair kernels), 2 / 3 / 4 = largest
// element count / extended node count / blob words
template <int DIM, bool FILL>
__global__ __launch_bounds__(TB_BS) void k_tb_build(const int2* __restrict__ pieces, const int32_t* __restrict__ order,
                                                    const int32_t* __restrict__ conn, const double* __restrict__ xyz,
                                                    const int32_t* __restrict__ n2e_ptr, const int32_t* __restrict__ n2e,
                                                    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, int dofs,
                                                    int full, TileHdr* __restrict__ hdr, int64_t* __restrict__ words,
                                                    int32_t* __restrict__ split, int32_t* __restrict__ counters,
                                                    uint32_t* __restrict__ blob) {
    constexpr int NEN = DIM + 1;
    __shared__ int32_t s_sort[TB_SORT];
    __shared__ int32_t s_els[TL_ELMAX];
    __shared__ int32_t s_ext[256];
    __shared__ int32_t s_nodes[TL_RMAX];
    __shared__ int32_t s_degp[TL_RMAX + 1];     // prefix of the nodes' adjacency lengths
    __shared__ int32_t s_nsl[TL_RMAX + 1];      // prefix of (slots + 1): the blob's sp
    __shared__ int32_t s_nbp[TL_RMAX + 1];      // prefix of the slots
    __shared__ int32_t s_scan[TB_BS + 1];
    __shared__ int32_t s_flag[2];
    __shared__ uint32_t s_elrec[FILL ? TL_ELMAX : 1];
    __shared__ uint16_t s_adj[FILL ? TB_SORT : 2];
    __shared__ uint8_t s_nbrl[FILL ? TL_RMAX * TB_NSMAX : 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_blob[FILL ? TL_BLOBMAX : 2];
    const int tid = threadIdx.x, tI = blockIdx.x;
    const int2 pc = pieces[tI];
    const int R = pc.y - pc.x;
    bool fail = R > TL_RMAX || R < 1;
    int total = 0, NS = 0, MAXS = 0, EL = 0, NE = 0;
    if (!fail) {
        if (tid < R) {
            const int32_t nd = order[pc.x + tid];
            s_nodes[tid] = nd;
            s_degp[tid + 1] = n2e_ptr[nd + 1] - n2e_ptr[nd];
            const int32_t row = nd * dofs, len = rowptr[row + 1] - rowptr[row];
            s_nbp[tid + 1] = full ? len / dofs : len;
        }
        __syncthreads();
        if (tid == 0) {
            s_degp[0] = s_nbp[0] = s_nsl[0] = 0;
            int mx = 0;
            for (int p = 0; p < R; ++p) {
                mx = max(mx, s_nbp[p + 1]);
                s_nsl[p + 1] = s_nsl[p] + s_nbp[p + 1] + 1;
                s_degp[p + 1] += s_degp[p];
                s_nbp[p + 1] += s_nbp[p];
            }
            s_flag[0] = mx;
        }
        __syncthreads();
        total = s_degp[R];
        NS = s_nsl[R];
        MAXS = s_flag[0];
        fail = total > TB_SORT || MAXS > TB_NSMAX;
    }
    if (!fail) {    // (uniform) the distinct elements of the tile, ascending
        int N = 2;
        while (N < total) N <<= 1;
        for (int i = tid; i < N; i += TB_BS) s_sort[i] = INT32_MAX;
        __syncthreads();
        for (int t = tid; t < total; t += TB_BS) {
            const int p = tb_upper(s_degp, R, t);
            s_sort[t] = n2e[n2e_ptr[s_nodes[p]] + (t - s_degp[p])] / NEN;
        }
        __syncthreads();
        tb_bitonic(s_sort, N, tid);
        const int C = (N + TB_BS - 1) / TB_BS, i0 = tid * C, i1 = min(N, i0 + C);
        int cnt = 0;
        for (int i = i0; i < i1; ++i) cnt += s_sort[i] != INT32_MAX && (i == 0 || s_sort[i] != s_sort[i - 1]) ? 1 : 0;
        int pos = tb_scan(cnt, s_scan, tid, &EL);
        for (int i = i0; i < i1; ++i)
            if (s_sort[i] != INT32_MAX && (i == 0 || s_sort[i] != s_sort[i - 1])) {
                if (pos < TL_ELMAX) s_els[pos] = s_sort[i];
                ++pos;
            }
        __syncthreads();
        fail = EL > TL_ELMAX;
    }
    if (!fail && s_nbp[R] > TB_SORT) fail = true;
    if (!fail) {    // the distinct vertices of those elements that are not nodes of the tile, ascending, behind the tile's nodes:
        // the union of the tile nodes' pattern rows (the neighbours of a node ARE the vertices of its elements; 405 entries to
        // sort for a 27-node tile of the Kuhn cube instead of the 1296 vertices of its 324 elements)
        const int nnb = s_nbp[R];
        int N = 2;
        while (N < nnb) N <<= 1;
        for (int i = tid; i < N; i += TB_BS) s_sort[i] = INT32_MAX;
        __syncthreads();
        for (int t = tid; t < nnb; t += TB_BS) {
            const int p = tb_upper(s_nbp, R, t), sl = t - s_nbp[p];
            s_sort[t] = colind[rowptr[s_nodes[p] * dofs] + (full ? sl * dofs : sl)] / dofs;
        }
        __syncthreads();
        tb_bitonic(s_sort, N, tid);
        const int C = (N + TB_BS - 1) / TB_BS, i0 = tid * C, i1 = min(N, i0 + C);
        int cnt = 0;
        for (int i = i0; i < i1; ++i)
            cnt += s_sort[i] != INT32_MAX && (i == 0 || s_sort[i] != s_sort[i - 1]) && tb_find(s_nodes, R, s_sort[i]) < 0 ? 1 : 0;
        int others = 0;
        int pos = tb_scan(cnt, s_scan, tid, &others);
        NE = R + others;
        for (int i = i0; i < i1; ++i)
            if (s_sort[i] != INT32_MAX && (i == 0 || s_sort[i] != s_sort[i - 1]) && tb_find(s_nodes, R, s_sort[i]) < 0) {
                if (R + pos < 256) s_ext[R + pos] = s_sort[i];
                ++pos;
            }
        if (tid < R) s_ext[tid] = s_nodes[tid];
        __syncthreads();
        fail = NE > TL_EXTMAX;
    }
    const int GN = total * NEN;
    const int nw_raw = 2 * DIM * NE + NE + R + EL + 2 * (R + 1) + ((NS + 1) >> 1) + ((GN + 1) >> 1);
    if (!fail) fail = nw_raw > TL_BLOBMAX || NS > 65535 || GN > 65535;
    if constexpr (!FILL) {
        if (tid == 0) {
            hdr[tI] = TileHdr{0u, (uint16_t)R, (uint16_t)NE, (uint16_t)EL, (uint16_t)NS, (uint16_t)GN, (uint16_t)MAXS};
            words[tI] = fail ? 0 : (int64_t)((nw_raw + 1) & ~1);
            split[tI] = fail ? 1 : 0;
            if (fail) {
                atomicAdd(&counters[0], 1);
                if (R <= 1) atomicMax(&counters[1], 1);
            } else {
                atomicMax(&counters[2], EL);
                atomicMax(&counters[3], NE);
                atomicMax(&counters[4], (nw_raw + 1) & ~1);
            }
        }
        return;
    } else {
        if (fail) {     // (cannot happen: the sizes pass accepted this tile)
            if (tid == 0) atomicMax(&counters[1], 1);
            return;
        }
        auto ext_local = [&](int32_t v) -> uint32_t {
            const int a = tb_find(s_nodes, R, v);
            return a >= 0 ? (uint32_t)a : (uint32_t)(R + tb_find(s_ext + R, NE - R, v));
        };
        const int o_ext = 2 * DIM * NE, o_nb = o_ext + NE, o_el = o_nb + R, o_gp = o_el + EL, o_sp = o_gp + R + 1, o_gs = o_sp + R + 1,
                  o_gl = o_gs + ((NS + 1) >> 1), nw = (nw_raw + 1) & ~1;
        for (int i = tid; i < nw; i += TB_BS) s_blob[i] = 0u;
        for (int q = tid; q < EL; q += TB_BS) {
            uint32_t rec = 0;
#pragma unroll
            for (int j = 0; j < NEN; ++j) rec |= ext_local(conn[(size_t)s_els[q] * NEN + j]) << (8 * j);
            s_elrec[q] = rec;
        }
        for (int t = tid; t < total; t += TB_BS) {
            const int p = tb_upper(s_degp, R, t);
            const int32_t idx = n2e[n2e_ptr[s_nodes[p]] + (t - s_degp[p])], el = idx / NEN, li = idx - el * NEN;
            s_adj[t] = (uint16_t)((tb_find(s_els, EL, el) << 2) | li);
        }
        const int nnb = s_nbp[R];
        for (int t = tid; t < nnb; t += TB_BS) {
            const int p = tb_upper(s_nbp, R, t), sl = t - s_nbp[p];
            const int32_t rs = rowptr[s_nodes[p] * dofs];
            s_nbrl[t] = (uint8_t)ext_local(colind[rs + (full ? sl * dofs : sl)] / dofs);
        }
        __syncthreads();
        double* cw = reinterpret_cast<double*>(s_blob);
        for (int t = tid; t < NE * DIM; t += TB_BS) cw[t] = xyz[(size_t)s_ext[t / DIM] * DIM + (t % DIM)];
        for (int t = tid; t < NE; t += TB_BS) s_blob[o_ext + t] = (uint32_t)s_ext[t];
        if (tid < R) {
            const int32_t rs = rowptr[s_nodes[tid] * dofs];
            s_blob[o_nb + tid] = (uint32_t)(dofs == 1 ? rs : (full ? rs / (dofs * dofs) : rs / dofs));
        }
        for (int t = tid; t < EL; t += TB_BS) s_blob[o_el + t] = s_elrec[t];
        if (tid <= R) {
            s_blob[o_gp + tid] = (uint32_t)(s_degp[tid] * NEN);
            s_blob[o_sp + tid] = (uint32_t)s_nsl[tid];
        }
        __syncthreads();
        // gather lists: a wave per node, a lane per slot; the lane walks the node's adjacency twice (count, fill)
        uint16_t* gs = reinterpret_cast<uint16_t*>(s_blob + o_gs);
        uint16_t* gl = reinterpret_cast<uint16_t*>(s_blob + o_gl);
        const int w = tid >> 6, lane = tid & 63;
        for (int p = w; p < R; p += TB_BS / 64) {
            const int nslot = s_nbp[p + 1] - s_nbp[p], qb = s_degp[p], qe = s_degp[p + 1];
            const uint32_t target = lane < nslot ? s_nbrl[s_nbp[p] + lane] : 0xffffu;
            int cnt = 0;
            for (int q = qb; q < qe; ++q) {
                const uint32_t rec = s_elrec[s_adj[q] >> 2];
                bool hit = false;
#pragma unroll
                for (int j = 0; j < NEN; ++j) hit = hit || ((rec >> (8 * j)) & 255u) == target;
                cnt += hit ? 1 : 0;
            }
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            int k = incl - cnt;
            if (lane < nslot) gs[s_nsl[p] + lane] = (uint16_t)k;
            if (lane == nslot - 1) gs[s_nsl[p] + nslot] = (uint16_t)incl;
            if (lane < nslot) {
                uint16_t* dst = gl + qb * NEN;
                for (int q = qb; q < qe; ++q) {
                    const uint32_t a = s_adj[q], rec = s_elrec[a >> 2];
#pragma unroll
                    for (int j = 0; j < NEN; ++j)
                        if (((rec >> (8 * j)) & 255u) == target) dst[k++] = (uint16_t)(((a >> 2) << 4) | ((a & 3u) << 2) | (uint32_t)j);
                }
            }
        }
        __syncthreads();
        uint32_t* out = blob + hdr[tI].off;
        for (int i = tid; i < nw; i += TB_BS) out[i] = s_blob[i];
    }
}

static int build_tiles_device(fedd_ctx* c) {
    const int dim = c->dim, nen = c->nen;
    const int64_t nn = c->n_own + c->n_rowg;          // nodes with rows
    c->tl_state = -1;
    if (nen != dim + 1 || nn <= 0 || c->n_elem <= 0 || !c->have_pattern || !c->have_adj || nn > ((int64_t)1 << 30)) return 0;
    hipStream_t st = c->stream;
    // ---- nodes -> cells of a coordinate lattice with ~27 nodes each (the host builder's lattice) ----
    double lo[3], hi[3];
    {
        DevBuf<double> mm;      // (the context's double scratch holds the quadrature tables of the assembly in progress)
        FEDD_TRY(mm.ensure(768));
        FEDD_TRY(bounding_box(c, nn, lo, hi, mm.p));
    }
    double V = 1.0;
    for (int d = 0; d < dim; ++d) V *= std::max(hi[d] - lo[d], 1e-300);
    const double target = dim == 3 ? 27.0 : 25.0;
    const double w = std::pow(V * target / (double)nn, 1.0 / dim);
    TbGeom gm;
    int64_t ncell = 1;
    for (int d = 0; d < 3; ++d) {
        gm.lo[d] = d < dim ? lo[d] : 0.0;
        gm.L[d] = d < dim ? hi[d] - lo[d] : 0.0;
        gm.g[d] = d < dim ? std::max(1, (int)std::floor((hi[d] - lo[d]) / w + 0.5)) : 1;
        ncell *= gm.g[d];
    }
    if (ncell > ((int64_t)1 << 30)) return 0;
    DevBuf<int32_t> cell, cptr, cursor, order, flag, split, cnt, counters;
    DevBuf<int64_t> words;
    DevBuf<int2> pieces[2];
    FEDD_TRY(cell.ensure((size_t)nn));
    FEDD_TRY(cptr.ensure((size_t)ncell + 2));
    FEDD_TRY(cursor.ensure((size_t)ncell + 2));
    FEDD_TRY(order.ensure((size_t)nn));
    FEDD_TRY(flag.ensure((size_t)ncell + 2));
    FEDD_TRY(counters.ensure(8));
    FEDD_HIP(hipMemsetAsync(cptr.p, 0, ((size_t)ncell + 2) * sizeof(int32_t), st));
    const dim3 blk(256), gnn((unsigned)((nn + 255) / 256)), gc((unsigned)((ncell + 255) / 256));
    if (dim == 3) hipLaunchKernelGGL(k_tb_cell<3>, gnn, blk, 0, st, (const double*)c->d_xyz.p, (int32_t)nn, gm, cell.p, cptr.p);
    else hipLaunchKernelGGL(k_tb_cell<2>, gnn, blk, 0, st, (const double*)c->d_xyz.p, (int32_t)nn, gm, cell.p, cptr.p);
    FEDD_TRY(exclusive_scan_i32(c, cptr.p, cptr.p, ncell, nullptr));
    FEDD_HIP(hipMemcpyAsync(cursor.p, cptr.p, ((size_t)ncell + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_tb_fill, gnn, blk, 0, st, (const int32_t*)cell.p, (int32_t)nn, cursor.p, order.p);
    hipLaunchKernelGGL(k_tb_sort_cells, dim3((unsigned)ncell), dim3(64), 0, st, (const int32_t*)cptr.p, (int32_t)ncell, order.p);
    hipLaunchKernelGGL(k_tb_flag_cells, gc, blk, 0, st, (const int32_t*)cptr.p, (int32_t)ncell, flag.p);
    int64_t npiece = 0;
    FEDD_TRY(exclusive_scan_i32(c, flag.p, flag.p, ncell, &npiece));
    if (npiece <= 0) return 0;
    FEDD_TRY(pieces[0].ensure((size_t)npiece));
    hipLaunchKernelGGL(k_tb_pieces0, gc, blk, 0, st, (const int32_t*)cptr.p, (int32_t)ncell, (const int32_t*)flag.p, pieces[0].p);
    // ---- sizes; tiles that do not fit are split in halves and sized again ----
    const int full = c->block_mode == FEDD_BLOCK_FULL ? 1 : 0;
    int cur = 0;
    int32_t h_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int round = 0;; ++round) {
        FEDD_CHECK(round < 40, "tile build: the splitting does not end");
        FEDD_TRY(c->tl_hdr.ensure((size_t)npiece * sizeof(TileHdr) / sizeof(uint32_t)));
        FEDD_TRY(words.ensure((size_t)npiece + 2));
        FEDD_TRY(split.ensure((size_t)npiece + 2));
        FEDD_HIP(hipMemsetAsync(counters.p, 0, 8 * sizeof(int32_t), st));
#define TB_BUILD(D, F)                                                                                                             \
    hipLaunchKernelGGL((k_tb_build<D, F>), dim3((unsigned)npiece), dim3(TB_BS), 0, st, (const int2*)pieces[cur].p,               \
                       (const int32_t*)order.p, (const int32_t*)c->d_conn.p, (const double*)c->d_xyz.p,                           \
                       (const int32_t*)c->d_n2e_ptr.p, (const int32_t*)c->d_n2e.p, (const int32_t*)c->d_rowptr.p,                 \
                       (const int32_t*)c->d_colind.p, c->dofs, full, reinterpret_cast<TileHdr*>(c->tl_hdr.p), words.p, split.p,   \
                       counters.p, c->tl_blob.p)
        if (dim == 3) TB_BUILD(3, false);
        else TB_BUILD(2, false);
        FEDD_HIP(hipMemcpyAsync(h_cnt, counters.p, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        FEDD_HIP(hipStreamSynchronize(st));
        if (h_cnt[1]) return 0;          // a single node that does not fit: the mesh stays on the pair kernels
        if (h_cnt[0] == 0) break;
        FEDD_TRY(cnt.ensure((size_t)npiece + 2));
        const dim3 gp((unsigned)((npiece + 255) / 256));
        hipLaunchKernelGGL(k_tb_split_count, gp, blk, 0, st, (const int32_t*)split.p, (int32_t)npiece, cnt.p);
        int64_t nnew = 0;
        FEDD_TRY(exclusive_scan_i32(c, cnt.p, cnt.p, npiece, &nnew));
        FEDD_TRY(pieces[cur ^ 1].ensure((size_t)nnew));
        hipLaunchKernelGGL(k_tb_split_scatter, gp, blk, 0, st, (const int2*)pieces[cur].p, (const int32_t*)split.p,
                           (const int32_t*)cnt.p, (int32_t)npiece, pieces[cur ^ 1].p);
        cur ^= 1;
        npiece = nnew;
    }
    // ---- blob offsets, then the blobs ----
    int64_t total_words = 0;
    FEDD_TRY(exclusive_scan_i64(c, words.p, words.p, npiece, &total_words));
    if (total_words >= ((int64_t)1 << 32)) return 0;      // 32-bit word offsets
    hipLaunchKernelGGL(k_tb_set_off, dim3((unsigned)((npiece + 255) / 256)), blk, 0, st, reinterpret_cast<TileHdr*>(c->tl_hdr.p),
                       (const int64_t*)words.p, (int32_t)npiece);
    FEDD_TRY(c->tl_blob.ensure((size_t)total_words + 2));
    FEDD_HIP(hipMemsetAsync(counters.p, 0, 8 * sizeof(int32_t), st));
    if (dim == 3) TB_BUILD(3, true);
    else TB_BUILD(2, true);
#undef TB_BUILD
    int32_t h_bad[2] = {0, 0};
    FEDD_HIP(hipMemcpyAsync(h_bad, counters.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    FEDD_HIP(hipStreamSynchronize(st));
    FEDD_HIP(hipGetLastError());
    if (h_bad[1]) return 0;
    c->tl_ntile = npiece;
    c->tl_max_el = h_cnt[2];
    c->tl_max_ext = h_cnt[3];
    c->tl_max_blob = h_cnt[4];
    c->tl_state = 1;
    return 0;
}

template <int DIM, int FORM>
int launch_tiles(fedd_ctx* c, const AsmArgs& a, int ntab) {
    if (c->tl_state == 0) {     // once per mesh; its wall time is kept for fedd_mesh_setup_info
        FEDD_HIP(hipStreamSynchronize(c->stream));
        const auto t0 = std::chrono::steady_clock::now();
        if (c->asm_tiles_host) FEDD_TRY(build_tiles(c));
        else FEDD_TRY(build_tiles_device(c));
        if (c->tl_state == 1) FEDD_TRY(build_tile_shapes(c));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        c->tl_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (c->tl_state != 1) return -1;
    constexpr int NEN = DIM + 1, PARK = FORM == F_LAPLACE ? NEN * NEN : NEN * DIM + 1;
    static_assert(sizeof(TileHdr) == 16, "tile header");
    const int lds_el = c->tl_max_el, lds_blob = (c->tl_max_blob + 1) & ~1;
    const size_t lds = ((size_t)ntab + 1 + (size_t)lds_el * PARK) * sizeof(double) + ((size_t)lds_blob + 2 * TL_RMAX + 2) * sizeof(uint32_t);
    if (lds > 96 * 1024 || lds_blob > TL_BLOBMAX) return -1;
    // Laplace: 448 lanes, the 324 elements and the 405 slots of a 3^3-node tile of the Kuhn cube each take one pass (4.27 -> 4.12 ms
    // at cfg 3); elasticity (9 items per slot: several passes anyway) is faster with 256 (94^3 cells: 2.86 against 3.34 ms)
    constexpr int BS = FORM == F_LAPLACE ? 448 : 256;
    // (the thresholded variant is its own instantiation: two compares and a select per element-matrix entry cost the
    // element phase 10 % -- 4.1 -> 4.5 ms at cfg 3 -- when they were compiled into the only one)
    auto kern = (FORM == F_LAPLACE && a.zero_eps > 0.0) ? k_assemble_tiles<DIM, FORM, BS, true> : k_assemble_tiles<DIM, FORM, BS, false>;
    if (lds > 64 * 1024) FEDD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // persistent workgroups: as many as fit the GPU at once (256 CUs x what the LDS allows), each a contiguous run of tiles
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
    const int64_t nwg = std::min<int64_t>(c->tl_ntile, (int64_t)256 * per_cu * (c->asm_u > 1 ? c->asm_u : 1));
    const int tiles_per_wg = (int)((c->tl_ntile + nwg - 1) / nwg);
    const int64_t grid = (c->tl_ntile + tiles_per_wg - 1) / tiles_per_wg;
    ScopedTimer tm(c, FEDD_T_ASSEMBLE);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BS), lds, c->stream, a, (const TileHdr*)c->tl_hdr.p,
                       (const uint32_t*)c->tl_blob.p, (const uint32_t*)c->tl_shape.p, (int32_t)c->tl_ntile, tiles_per_wg, c->block_mode, lds_el, lds_blob, c->asm_dbg);
    tm.stop();
    if (c->asm_dbg & 64) fprintf(stderr, "[tiles] %lld tiles (%d read the shape of an earlier one), max elements %d, blob words %d, LDS %zu bytes, %lld workgroups x %d tiles\n", (long long)c->tl_ntile, c->tl_nshared, lds_el, lds_blob, lds, (long long)grid, tiles_per_wg);
    FEDD_HIP(hipGetLastError());
    return 0;
}

// asm_kind 0: slot-addressed kernel, falling back to the pair-parallel sweep where it does not fit; 2: the sweep
template <int DIM, int NEN, int FORM>
int launch_matrix(fedd_ctx* c, const AsmArgs& a, int ntab, int64_t n_rows, int rowcap) {
    // Measured (one MI355X): elasticity, 94^3 cells, FULL blocks: sweep 7.9 ms, slots 5.5 ms; B / B^T likewise;
    // Laplace, 214^3 cells: sweep 5.15 ms, slots 5.5-5.7 ms (the one-wave accumulation of phase 2 costs what the sweep
    // over 96 contributions costs).  asm_kind 0 picks by the contributions per pair, 3 forces the slot kernel.
    constexpr bool many = PairCfg<DIM, NEN, FORM>::CPP > NEN;
    if (((c->asm_kind == 0 && many) || c->asm_kind == 3) && n_rows > 0) {
        const int rc = launch_slots<DIM, NEN, FORM>(c, a, ntab, n_rows, rowcap);
        if (rc >= 0) return rc;
    }
    return launch_pairs<DIM, NEN, FORM>(c, a, ntab, n_rows, rowcap);
}

// ---------------------------------------------------------------------------------------------
// Row sums of the P2 scalar forms from the element matrices of k_elem_matrix, by GATHER LISTS built once per mesh
// (k_p2_lists): for every node-level nonzero (node p, slot s) the (adjacency entry q, local column j) pairs that contribute
// to it, in adjacency order -- 16 bits each, q << 4 | j, behind a 16-bit start per nonzero; a node's list starts at
// NEN x its adjacency start, no scan needed.  k_p2_gather: a wave per node, a lane per slot, every lane adds its few (2.6 on
// average) element-matrix entries in list order and writes its CSR slot(s): no search, no atomics, bitwise reproducible, and
// the summation order of the pair kernels.  (The pair kernels spend their time searching the row for every one of the 157 M
// contributions of a 64^3-cell P2 cube: 11.4 ms with or without the element matrices.)
// ---------------------------------------------------------------------------------------------
template <int NEN>
__global__ __launch_bounds__(256) void k_p2_lists(const int32_t* __restrict__ conn, const int32_t* __restrict__ n2e_ptr,
                                                  const int32_t* __restrict__ n2e, const int32_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ colind, int dofs, int full, int32_t nn,
                                                  uint16_t* __restrict__ soff, uint16_t* __restrict__ src, int32_t* __restrict__ bad) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t p = (int64_t)blockIdx.x * 4 + w; p < nn; p += (int64_t)gridDim.x * 4) {
        const int32_t ab = n2e_ptr[p], deg = n2e_ptr[p + 1] - ab;
        const int32_t row = (int32_t)p * dofs, rs = rowptr[row], len = rowptr[row + 1] - rs;
        const int nslot = full ? len / dofs : len, step = full ? dofs : 1;
        const int32_t nbn = dofs == 1 ? rs : (full ? rs / (dofs * dofs) : rs / dofs);
        const int64_t base = (int64_t)ab * NEN;
        if (deg > 4095 || deg * NEN > 65535) {
            if (lane == 0) atomicMax(bad, 1);
            continue;
        }
        int running = 0;
        for (int s0 = 0; s0 < nslot; s0 += 64) {
            const int sl = s0 + lane;
            const bool on = sl < nslot;
            const int32_t v = on ? colind[rs + sl * step] / dofs : -1;
            int cnt = 0;
            for (int q = 0; q < deg; ++q) {
                const int32_t* __restrict__ en = conn + (int64_t)(n2e[ab + q] / NEN) * NEN;
#pragma unroll
                for (int j = 0; j < NEN; ++j) cnt += en[j] == v ? 1 : 0;
            }
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            int k = running + incl - cnt;
            if (on) soff[nbn + sl] = (uint16_t)k;
            if (on)
                for (int q = 0; q < deg; ++q) {
                    const int32_t* __restrict__ en = conn + (int64_t)(n2e[ab + q] / NEN) * NEN;
#pragma unroll
                    for (int j = 0; j < NEN; ++j)
                        if (en[j] == v) src[base + k++] = (uint16_t)((q << 4) | j);
                }
            running += __shfl(incl, 63, 64);
        }
    }
}

template <int NEN>
__global__ __launch_bounds__(256) void k_p2_gather(const int32_t* __restrict__ n2e_ptr, const int32_t* __restrict__ n2e,
                                                   const int32_t* __restrict__ rowptr, int dofs, int32_t nn,
                                                   const uint16_t* __restrict__ soff, const uint16_t* __restrict__ src,
                                                   const double* __restrict__ ke, double* __restrict__ val) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t p = (int64_t)blockIdx.x * 4 + w; p < nn; p += (int64_t)gridDim.x * 4) {
        const int32_t ab = n2e_ptr[p], deg = n2e_ptr[p + 1] - ab;
        const int32_t row = (int32_t)p * dofs, rs = rowptr[row], nslot = rowptr[row + 1] - rs;     // (scalar or diagonal blocks)
        const int32_t nbn = dofs == 1 ? rs : rs / dofs;
        const uint16_t* __restrict__ sp = src + (int64_t)ab * NEN;
        const int total = deg * NEN;
        for (int sl = lane; sl < nslot; sl += 64) {
            const int b = soff[nbn + sl], e2 = sl + 1 < nslot ? (int)soff[nbn + sl + 1] : total;
            double acc = 0.0;
            for (int k = b; k < e2; ++k) {
                const uint32_t sr = sp[k];
                acc += ke[(int64_t)n2e[ab + (sr >> 4)] * NEN + (sr & 15u)];
            }
            for (int comp = 0; comp < dofs; ++comp) val[rowptr[row + comp] + sl] = acc;
        }
    }
}

// the gather lists of the current mesh (built at the first P2 assembly that uses them); c->p2_state = -1: not applicable
template <int NEN>
int p2_lists_build(fedd_ctx* c) {
    c->p2_state = -1;
    const int64_t nn = c->n_own + c->n_rowg;
    if (!c->have_pattern || !c->have_adj || c->block_mode == FEDD_BLOCK_FULL || nn <= 0) return 0;
    int32_t n2e_total = 0;
    FEDD_HIP(hipMemcpyAsync(&n2e_total, c->d_n2e_ptr.p + nn, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    const int64_t node_nnz = c->nnz_ext / c->dofs;      // scalar: nnz; diagonal blocks: dofs rows of the node-level length each
    FEDD_TRY(c->d_p2_soff.ensure((size_t)node_nnz + 2));
    FEDD_TRY(c->d_p2_src.ensure((size_t)n2e_total * NEN + 2));
    FEDD_TRY(c->d_flags.ensure(16));
    int32_t* bad = c->d_flags.p + 6;
    FEDD_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), c->stream));
    const int nwg = (int)std::min<int64_t>((nn + 3) / 4, 256 * 32);
    hipLaunchKernelGGL(k_p2_lists<NEN>, dim3((unsigned)nwg), dim3(256), 0, c->stream, (const int32_t*)c->d_conn.p,
                       (const int32_t*)c->d_n2e_ptr.p, (const int32_t*)c->d_n2e.p, (const int32_t*)c->d_rowptr.p,
                       (const int32_t*)c->d_colind.p, c->dofs, 0, (int32_t)nn, c->d_p2_soff.p, c->d_p2_src.p, bad);
    int32_t h_bad = 0;
    FEDD_HIP(hipMemcpyAsync(&h_bad, bad, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    FEDD_HIP(hipGetLastError());
    if (!h_bad) c->p2_state = 1;
    return 0;
}

template <int DIM, int NEN>
int launch_assemble(fedd_ctx* c, int kform, const AsmArgs& a, int ntab) {
    if constexpr (NEN == DIM + 1) {
        // element-major tiles for the P1 forms they cover (asm_kind 4; 0: by default for those forms), else the pair kernels
        if ((c->asm_kind == 4 || (c->asm_kind == 0 && c->asm_tiles)) && a.nq == 1 && (kform == F_LAPLACE || kform == F_LINELAS)) {
            const int rc = kform == F_LAPLACE ? launch_tiles<DIM, F_LAPLACE>(c, a, ntab) : launch_tiles<DIM, F_LINELAS>(c, a, ntab);
            if (rc >= 0) return rc;
        }
    }
    if constexpr (NEN > DIM + 1) {
        // P2: the element matrices once per element (k_elem_matrix), the rows summed from them (option "asm_p2_elem" 0: the pair
        // kernels re-derive the row of every (row, element) pair)
        if (c->asm_p2_elem && c->asm_kind != 1 && (kform == F_LAPLACE || kform == F_MASS)) {
            AsmArgs ae = a;
            if (kform == F_LAPLACE) FEDD_TRY((launch_elem_matrices<DIM, NEN, F_LAPLACE>(c, ae, ntab)));
            else FEDD_TRY((launch_elem_matrices<DIM, NEN, F_MASS>(c, ae, ntab)));
            // rows from the gather lists (scalar rows or diagonal blocks; "asm_p2_elem" 2: through the pair kernels)
            if (c->asm_p2_elem == 1 && c->block_mode != FEDD_BLOCK_FULL) {
                if (c->p2_state == 0) {
                    FEDD_HIP(hipStreamSynchronize(c->stream));
                    const auto t0 = std::chrono::steady_clock::now();
                    FEDD_TRY(p2_lists_build<NEN>(c));
                    FEDD_HIP(hipStreamSynchronize(c->stream));
                    c->tl_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                }
                if (c->p2_state == 1) {
                    const int64_t nn = c->n_own + c->n_rowg;
                    const int nwg = (int)std::min<int64_t>((nn + 3) / 4, 256 * 32);
                    ScopedTimer t(c, FEDD_T_ASSEMBLE);
                    hipLaunchKernelGGL(k_p2_gather<NEN>, dim3((unsigned)nwg), dim3(256), 0, c->stream, (const int32_t*)c->d_n2e_ptr.p,
                                       (const int32_t*)c->d_n2e.p, (const int32_t*)c->d_rowptr.p, c->dofs, (int32_t)nn,
                                       (const uint16_t*)c->d_p2_soff.p, (const uint16_t*)c->d_p2_src.p, (const double*)c->d_ke.p,
                                       c->d_val.p);
                    t.stop();
                    FEDD_HIP(hipGetLastError());
                    return 0;
                }
            }
            if (kform == F_LAPLACE) return launch_matrix<DIM, NEN, F_LAPLACE>(c, ae, ntab, c->n_rows_ext, c->max_row_nnz);
            return launch_matrix<DIM, NEN, F_MASS>(c, ae, ntab, c->n_rows_ext, c->max_row_nnz);
        }
    }
    if (c->asm_kind != 1) {
        if (kform == F_LAPLACE) return launch_matrix<DIM, NEN, F_LAPLACE>(c, a, ntab, c->n_rows_ext, c->max_row_nnz);
        if (kform == F_MASS) return launch_matrix<DIM, NEN, F_MASS>(c, a, ntab, c->n_rows_ext, c->max_row_nnz);
        return launch_matrix<DIM, NEN, F_LINELAS>(c, a, ntab, c->n_rows_ext, c->max_row_nnz);
    }
    if (false) {
        if (kform == F_LAPLACE) return launch_pairs<DIM, NEN, F_LAPLACE>(c, a, ntab, c->n_rows_ext, c->max_row_nnz);
        if (kform == F_MASS) return launch_pairs<DIM, NEN, F_MASS>(c, a, ntab, c->n_rows_ext, c->max_row_nnz);
        return launch_pairs<DIM, NEN, F_LINELAS>(c, a, ntab, c->n_rows_ext, c->max_row_nnz);
    }
    const int rowcap = std::max(1, c->max_row_nnz);
    int bs = 256;
    auto need = [&](int b) { return ((size_t)ntab + (size_t)rowcap * b) * sizeof(double); };
    while (bs > 64 && need(bs) > 64 * 1024) bs >>= 1;
    const size_t lds = need(bs);
    FEDD_CHECK(lds <= 160 * 1024, "assembly: a CSR row with %d entries does not fit the LDS row buffer", rowcap);
    const dim3 grid((unsigned)((c->n_rows_ext + bs - 1) / bs)), block(bs);
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            FEDD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ScopedTimer t(c, FEDD_T_ASSEMBLE);
        hipLaunchKernelGGL(kern, grid, block, lds, c->stream, a);
        t.stop();
        FEDD_HIP(hipGetLastError());
        return 0;
    };
    if (kform == F_LAPLACE) return go(k_assemble<DIM, NEN, F_LAPLACE>);
    if (kform == F_MASS) return go(k_assemble<DIM, NEN, F_MASS>);
    return go(k_assemble<DIM, NEN, F_LINELAS>);
}

// quadrature weights, basis values / gradients of the mesh's element and the P1 (pressure) basis at
// the same points -> d_dtmp0, layout w | phi | dphi | psi
int upload_tables(fedd_ctx* c, int degree, int& nq, int& ntab) {
    const int dim = c->dim, nen = c->nen;
    FeTables tb, tp;
    FEDD_TRY(fe_tables(dim, nen, degree, tb));
    FEDD_TRY(fe_tables(dim, dim + 1, degree, tp));
    nq = tb.nq;
    ntab = nq * (1 + nen + nen * dim + dim + 1);
    std::vector<double> host(ntab);
    std::copy(tb.w.begin(), tb.w.end(), host.begin());
    std::copy(tb.phi.begin(), tb.phi.end(), host.begin() + nq);
    std::copy(tb.dphi.begin(), tb.dphi.end(), host.begin() + nq + nq * nen);
    std::copy(tp.phi.begin(), tp.phi.end(), host.begin() + nq + nq * nen + nq * nen * dim);
    FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>(ntab, c->d_dtmp0.cap)));
    FEDD_HIP(hipMemcpyAsync(c->d_dtmp0.p, host.data(), ntab * sizeof(double), hipMemcpyHostToDevice, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));  // `host` is a local
    return 0;
}

// B pattern from the scalar node pattern: pressure node i (< n_p) couples to all dim components of
// every velocity node of its elements.
__global__ void k_div_pattern(const int32_t* __restrict__ nptr, const int32_t* __restrict__ ncol, int32_t n_p, int dim,
                              int32_t* __restrict__ rowptr, int32_t* __restrict__ colind) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_p) return;
    rowptr[i] = nptr[i] * dim;
    if (i == n_p) return;
    const int32_t b = nptr[i], nn = nptr[i + 1] - b;
    for (int32_t s = 0; s < nn; ++s)
        for (int d = 0; d < dim; ++d) colind[(b + s) * dim + d] = ncol[b + s] * dim + d;
}

// B^T: velocity node j couples to the vertices (< n_p) of its elements
__global__ void k_divt_count(const int32_t* __restrict__ nptr, const int32_t* __restrict__ ncol, int32_t n_v, int32_t n_p,
                             int32_t* __restrict__ cnt) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_v) return;
    int32_t k = 0;
    for (int32_t p = nptr[j]; p < nptr[j + 1]; ++p) k += ncol[p] < n_p ? 1 : 0;
    cnt[j] = k;
}

__global__ void k_divt_fill(const int32_t* __restrict__ nptr, const int32_t* __restrict__ ncol, int32_t n_v, int32_t n_p,
                            int dim, const int32_t* __restrict__ cptr, int32_t* __restrict__ rowptr,
                            int32_t* __restrict__ colind) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_v) return;
    if (j == n_v) {
        rowptr[(int64_t)n_v * dim] = cptr[n_v] * dim;
        return;
    }
    const int32_t nn = cptr[j + 1] - cptr[j];
    for (int d = 0; d < dim; ++d) {
        const int32_t start = cptr[j] * dim + d * nn;
        rowptr[(int64_t)j * dim + d] = start;
        int32_t k = 0;
        for (int32_t p = nptr[j]; p < nptr[j + 1]; ++p)
            if (ncol[p] < n_p) colind[start + k++] = ncol[p];
    }
}

}  // namespace

// FE::assemblyDivAndDivT (feddlib/core/FE/FE_def.hpp:1932-2057) for velocity = the mesh's element
// (P2 or P1) and pressure = P1 on the vertices; the P1 nodes are the first n_p node ids (that is how
// the P2 mesh is built from the P1 mesh).  Unscaled; Stokes::assemble applies the -1 afterwards.
// Uses (and overwrites) the system slot for the scalar node pattern, so blocks that must survive
// have to be stored first.
int assemble_div(fedd_ctx* c, int64_t n_p, int slot_b, int slot_bt) {
    c->cs_valid = false;   // the solver's compacted SpMV stream follows the matrix values
    FEDD_CHECK(c->nranks == 1, "assemblyDivAndDivT: one rank only for now");
    FEDD_CHECK(n_p > 0 && n_p <= c->n_own, "assemblyDivAndDivT: %lld pressure nodes of %lld nodes", (long long)n_p, (long long)c->n_own);
    const int dim = c->dim, nen = c->nen;
    const int32_t n_v = (int32_t)c->n_own;
    if (!c->have_adj) FEDD_TRY(build_adjacency(c));
    FEDD_TRY(build_pattern(c, 1, FEDD_BLOCK_SCALAR));  // scalar node pattern -> system slot
    const int32_t* nptr = c->d_rowptr.p;
    const int32_t* ncol = c->d_colind.p;
    const int node_rowcap = c->max_row_nnz;
    DevCsr& B = c->aux[slot_b];
    DevCsr& BT = c->aux[slot_bt];
    // ---- patterns ----
    int32_t h_np = 0;
    FEDD_HIP(hipMemcpyAsync(&h_np, nptr + n_p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    B.n_rows = n_p; B.n_cols = (int64_t)n_v * dim; B.nnz = (int64_t)h_np * dim; B.max_row_nnz = node_rowcap * dim;
    FEDD_TRY(B.rowptr.ensure((size_t)n_p + 1));
    FEDD_TRY(B.colind.ensure((size_t)B.nnz));
    FEDD_TRY(B.val.ensure((size_t)B.nnz));
    const dim3 blk(256);
    hipLaunchKernelGGL(k_div_pattern, dim3((unsigned)((n_p + 1 + 255) / 256)), blk, 0, c->stream, nptr, ncol, (int32_t)n_p, dim,
                       B.rowptr.p, B.colind.p);
    FEDD_TRY(c->d_itmp1.ensure(std::max<size_t>((size_t)n_v + 1, c->d_itmp1.cap)));
    int32_t* cptr = c->d_itmp1.p;
    hipLaunchKernelGGL(k_divt_count, dim3((unsigned)((n_v + 255) / 256)), blk, 0, c->stream, nptr, ncol, n_v, (int32_t)n_p, cptr);
    int32_t mx = 0;
    FEDD_TRY(reduce_max_i32(c, cptr, n_v, &mx));
    int64_t tot = 0;
    FEDD_TRY(exclusive_scan_i32(c, cptr, cptr, n_v, &tot));
    BT.n_rows = (int64_t)n_v * dim; BT.n_cols = n_p; BT.nnz = tot * dim; BT.max_row_nnz = mx;
    FEDD_TRY(BT.rowptr.ensure((size_t)BT.n_rows + 1));
    FEDD_TRY(BT.colind.ensure((size_t)BT.nnz));
    FEDD_TRY(BT.val.ensure((size_t)BT.nnz));
    hipLaunchKernelGGL(k_divt_fill, dim3((unsigned)((n_v + 1 + 255) / 256)), blk, 0, c->stream, nptr, ncol, n_v, (int32_t)n_p, dim,
                       (const int32_t*)cptr, BT.rowptr.p, BT.colind.p);
    FEDD_HIP(hipGetLastError());
    // ---- values: determineDegree(dim, FE1, FE2, Grad, Std) (FE_def.hpp:1962) ----
    int degree = fe_degree(nen, dim, true) + 1;
    if (degree == 0) degree = 1;
    int nq = 0, ntab = 0;
    FEDD_TRY(upload_tables(c, degree, nq, ntab));
    AsmArgs a;
    a.conn = c->d_conn.p; a.n2e_ptr = c->d_n2e_ptr.p; a.n2e = c->d_n2e.p; a.xyz = c->d_xyz.p; a.tab = c->d_dtmp0.p;
    a.nq = nq; a.p0 = a.p1 = 0.0;
    a.ke = nullptr;
    a.zero_eps = c->asm_zero_eps;       // doSetZeros: B and B^T threshold their element contributions (FE_def.hpp:2002-2004, 2032-2034)
    AsmArgs ab = a, at = a;
    ab.rowptr = B.rowptr.p; ab.colind = B.colind.p; ab.val = B.val.p; ab.n_rows = (int32_t)n_p; ab.dofs = 1;
    at.rowptr = BT.rowptr.p; at.colind = BT.colind.p; at.val = BT.val.p; at.n_rows = (int32_t)BT.n_rows; at.dofs = dim;
#define DIV_LAUNCH(D, N)                                                                               \
    do {                                                                                               \
        FEDD_TRY((launch_matrix<D, N, F_DIV>(c, ab, ntab, n_p, B.max_row_nnz)));                       \
        FEDD_TRY((launch_matrix<D, N, F_DIVT>(c, at, ntab, BT.n_rows, BT.max_row_nnz)));               \
    } while (0)
    if (dim == 2 && nen == 3) DIV_LAUNCH(2, 3);
    else if (dim == 2 && nen == 6) DIV_LAUNCH(2, 6);
    else if (dim == 3 && nen == 4) DIV_LAUNCH(3, 4);
    else DIV_LAUNCH(3, 10);
#undef DIV_LAUNCH
    B.valid = BT.valid = true;
    c->have_pattern = false;  // the system slot only holds the scratch node pattern now
    c->have_schwarz = false;
    return 0;
}

namespace {
}

int assemble_matrix(fedd_ctx* c, int form, const double* params) {
    c->cs_valid = false;   // the solver's compacted SpMV stream follows the matrix values
    const int dim = c->dim, nen = c->nen;
    int kform, degree;
    const int dg = fe_degree(nen, dim, true), ds = fe_degree(nen, dim, false);
    switch (form) {
        case FEDD_FORM_LAPLACE:
            FEDD_CHECK(c->dofs == 1, "assemblyLaplace needs a scalar pattern");
            kform = F_LAPLACE; degree = dg + dg; break;
        case FEDD_FORM_LAPLACE_VEC:
            FEDD_CHECK(c->dofs == dim && c->block_mode == FEDD_BLOCK_DIAG, "assemblyLaplaceVecField needs a DIAG pattern with dim dofs per node");
            kform = F_LAPLACE; degree = dg + dg; break;
        case FEDD_FORM_MASS:
            FEDD_CHECK(c->dofs == 1, "assemblyMass(Scalar) needs a scalar pattern");
            kform = F_MASS; degree = ds + ds; break;
        case FEDD_FORM_MASS_VEC:
            FEDD_CHECK(c->dofs == dim && c->block_mode == FEDD_BLOCK_DIAG, "assemblyMass(Vector) needs a DIAG pattern with dim dofs per node");
            kform = F_MASS; degree = ds + ds; break;
        case FEDD_FORM_BDSTAB:
            FEDD_CHECK(c->dofs == 1, "assemblyBDStabilization needs a scalar pattern");
            FEDD_CHECK(nen == dim + 1, "assemblyBDStabilization: only implemented for P1 (FE_def.hpp:2156)");
            kform = F_MASS; degree = ds + ds; break;
        case FEDD_FORM_LINELAS:
            FEDD_CHECK(c->dofs == dim && c->block_mode == FEDD_BLOCK_FULL, "assemblyLinElasXDim needs a FULL pattern with dim dofs per node");
            FEDD_CHECK(params, "assemblyLinElasXDim needs params = {lambda, mu}");
            kform = F_LINELAS; degree = dg + dg; break;
        default:
            FEDD_CHECK(false, "fedd_assemble: unknown form %d", form);
    }
    if (degree == 0) degree = 1;  // FE::determineDegree, FE_def.hpp:5508-5509
    int nq = 0, ntab = 0;
    FEDD_TRY(upload_tables(c, degree, nq, ntab));
    AsmArgs a;
    a.conn = c->d_conn.p; a.n2e_ptr = c->d_n2e_ptr.p; a.n2e = c->d_n2e.p; a.rowptr = c->d_rowptr.p;
    a.colind = c->d_colind.p; a.xyz = c->d_xyz.p; a.val = c->d_val.p; a.tab = c->d_dtmp0.p;
    a.nq = nq; a.n_rows = (int32_t)c->n_rows_ext; a.dofs = c->dofs;
    a.p0 = params ? params[0] : 0.0;
    a.p1 = params ? params[1] : 0.0;
    a.ke = nullptr;
    // doSetZeros: of the matrix forms built here only the vector Laplacian thresholds (FE_def.hpp:719-721; assemblyLaplace does not)
    a.zero_eps = form == FEDD_FORM_LAPLACE_VEC ? c->asm_zero_eps : 0.0;
    if (kform == F_MASS) {   // the constant the Bochev-Dohrmann block takes off every mass entry: |ref. element| x scale (FE_def.hpp:2183-2192)
        a.p0 = form == FEDD_FORM_BDSTAB ? (dim == 2 ? 0.5 : 1.0 / 6.0) : 0.0;
        a.p1 = form == FEDD_FORM_BDSTAB ? (dim == 2 ? 1.0 / 9.0 : 1.0 / 16.0) : 0.0;
    }
    c->have_schwarz = false;
    if (dim == 2 && nen == 3) return launch_assemble<2, 3>(c, kform, a, ntab);
    if (dim == 2 && nen == 6) return launch_assemble<2, 6>(c, kform, a, ntab);
    if (dim == 3 && nen == 4) return launch_assemble<3, 4>(c, kform, a, ntab);
    return launch_assemble<3, 10>(c, kform, a, ntab);
}

int assemble_rhs(fedd_ctx* c, int dofs, const double* f_const, int extra_degree) {
    const int dim = c->dim, nen = c->nen;
    int degree = fe_degree(nen, dim, false);
    if (degree == 0) degree = 1;
    degree += extra_degree;  // FE_def.hpp:4717-4718
    FeTables tb;
    FEDD_TRY(fe_tables(dim, nen, degree, tb));
    RhsArgs a;
    a.conn = c->d_conn.p; a.n2e_ptr = c->d_n2e_ptr.p; a.n2e = c->d_n2e.p; a.xyz = c->d_xyz.p;
    a.rhs = c->d_rhs.p; a.n_own = (int32_t)c->n_own; a.nen = nen; a.dofs = dofs;
    for (int i = 0; i < 10; ++i) a.base[i] = 0.0;
    for (int i = 0; i < nen; ++i) {
        double s = 0.0;
        for (int q = 0; q < tb.nq; ++q) s += tb.w[q] * tb.phi[(size_t)q * nen + i];
        a.base[i] = s;
    }
    for (int d = 0; d < MAX_DOFS; ++d) a.f[d] = d < dofs ? f_const[d] : 0.0;
    const dim3 grid((unsigned)((c->n_own + 255) / 256)), block(256);
    FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>((size_t)c->n_elem, c->d_dtmp0.cap)));
    const dim3 egrid((unsigned)((c->n_elem + 255) / 256));
    ScopedTimer t(c, FEDD_T_RHS);
    if (dim == 2)
        hipLaunchKernelGGL(k_elem_absdet<2>, egrid, block, 0, c->stream, (const int32_t*)c->d_conn.p, nen,
                           (const double*)c->d_xyz.p, c->n_elem, c->d_dtmp0.p);
    else
        hipLaunchKernelGGL(k_elem_absdet<3>, egrid, block, 0, c->stream, (const int32_t*)c->d_conn.p, nen,
                           (const double*)c->d_xyz.p, c->n_elem, c->d_dtmp0.p);
    const int cap = std::max(256, std::min(RHS_NPB * std::max(1, c->max_deg), 6144));
    hipLaunchKernelGGL(k_rhs, dim3((unsigned)((c->n_own + RHS_NPB - 1) / RHS_NPB)), block, (size_t)cap * sizeof(double),
                       c->stream, a, (const double*)c->d_dtmp0.p, cap);
    t.stop();
    FEDD_HIP(hipGetLastError());
    return 0;
}

int apply_dirichlet_nodes(fedd_ctx* c, int64_t n, const int32_t* nodes, const int32_t* comp_mask, const double* values) {
    c->cs_valid = false;   // the solver's compacted SpMV stream follows the matrix values
    if (n == 0) return 0;
    const int dofs = c->dofs;
    for (int64_t k = 0; k < n; ++k)
        FEDD_CHECK(nodes[k] >= 0 && nodes[k] < c->n_own + c->n_rowg, "fedd_dirichlet_nodes: node %d has no rows on this rank", nodes[k]);
    FEDD_TRY(c->d_itmp0.ensure(std::max<size_t>((size_t)n * (1 + dofs), c->d_itmp0.cap)));
    FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>((size_t)n * dofs, c->d_dtmp0.cap)));
    int32_t* d_nodes = c->d_itmp0.p;
    int32_t* d_mask = comp_mask ? c->d_itmp0.p + n : nullptr;
    FEDD_HIP(hipMemcpyAsync(d_nodes, nodes, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    if (comp_mask) FEDD_HIP(hipMemcpyAsync(d_mask, comp_mask, (size_t)n * dofs * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    FEDD_HIP(hipMemcpyAsync(c->d_dtmp0.p, values, (size_t)n * dofs * sizeof(double), hipMemcpyHostToDevice, c->stream));
    ScopedTimer t(c, FEDD_T_DIRICHLET);
    hipLaunchKernelGGL(k_dirichlet_nodes, dim3((unsigned)((n * dofs + 255) / 256)), dim3(256), 0, c->stream,
                       (const int32_t*)d_nodes, (const int32_t*)d_mask, (const double*)c->d_dtmp0.p, (int32_t)n, dofs,
                       (const int32_t*)c->d_rowptr.p, (const int32_t*)c->d_colind.p, c->d_val.p, c->d_rhs.p, c->d_isdir.p);
    t.stop();
    FEDD_HIP(hipGetLastError());
    FEDD_HIP(hipStreamSynchronize(c->stream));  // host staging buffers are the caller's
    c->have_schwarz = false;
    return 0;
}

// generic variant on system rows (merged block systems): row <- unit row, rhs <- value.  On a merged
// matrix this equals setLocalRowOne on the diagonal block + setLocalRowZero on the off-diagonal
// blocks of that block row (BCBuilder_def.hpp:589-707) applied before the merge.
int apply_dirichlet_rows(fedd_ctx* c, int64_t n, const int32_t* rows, const double* values) {
    c->cs_valid = false;   // the solver's compacted SpMV stream follows the matrix values
    if (n == 0) return 0;
    for (int64_t k = 0; k < n; ++k)
        FEDD_CHECK(rows[k] >= 0 && rows[k] < c->n_rows, "fedd_dirichlet_rows: row %d out of range", rows[k]);
    FEDD_TRY(c->d_itmp0.ensure(std::max<size_t>((size_t)n, c->d_itmp0.cap)));
    FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>((size_t)n, c->d_dtmp0.cap)));
    FEDD_HIP(hipMemcpyAsync(c->d_itmp0.p, rows, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    FEDD_HIP(hipMemcpyAsync(c->d_dtmp0.p, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // reuse the node kernel with one dof per "node": row ids are the node ids
    ScopedTimer t(c, FEDD_T_DIRICHLET);
    hipLaunchKernelGGL(k_dirichlet_nodes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       (const int32_t*)c->d_itmp0.p, (const int32_t*)nullptr, (const double*)c->d_dtmp0.p, (int32_t)n, 1,
                       (const int32_t*)c->d_rowptr.p, (const int32_t*)c->d_colind.p, c->d_val.p, c->d_rhs.p, c->d_isdir.p);
    t.stop();
    FEDD_HIP(hipGetLastError());
    FEDD_HIP(hipStreamSynchronize(c->stream));
    c->have_schwarz = false;
    return 0;
}

int apply_dirichlet(fedd_ctx* c, int n_bc, const int32_t* flags, const int32_t* comp_mask, const double* values) {
    c->cs_valid = false;   // the solver's compacted SpMV stream follows the matrix values
    BcArgs b;
    b.n = n_bc;
    b.dofs = c->dofs;
    for (int k = 0; k < n_bc; ++k) {
        b.flag[k] = flags[k];
        for (int d = 0; d < c->dofs; ++d) {
            b.mask[k * c->dofs + d] = comp_mask ? comp_mask[k * c->dofs + d] : 1;
            b.value[k * c->dofs + d] = values[k * c->dofs + d];
        }
    }
    // row-ghost rows get the same treatment as owned ones (their flags follow the owned flags in d_flag)
    const dim3 grid((unsigned)((c->n_rows_ext + 255) / 256)), block(256);
    ScopedTimer t(c, FEDD_T_DIRICHLET);
    hipLaunchKernelGGL(k_dirichlet, grid, block, 0, c->stream, b, c->d_flag.p, c->d_rowptr.p, c->d_colind.p,
                       c->d_val.p, c->d_rhs.p, c->d_isdir.p, (int32_t)c->n_rows_ext);
    t.stop();
    FEDD_HIP(hipGetLastError());
    c->have_schwarz = false;
    return 0;
}

}  // namespace fedd
