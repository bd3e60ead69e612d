// Batched exact local inverses for the Schwarz subdomains on the f64 matrix cores.
//
// Same job as schwarz.hip k_invert_reg (the "local dense factor per overlapping subdomain" of the
// one-level operator the reference gets from FROSch's AlgebraicOverlappingOperator with an exact
// local solver, feddlib/problems/tests/laplace/parametersPrec.xml:33-61), for plain (not merged
// saddle-point) systems with n <= 128 dofs: Gauss-Jordan without row exchanges, but swept in blocks
// of four pivots, so that the update is a rank-4 product on v_mfma_f64_16x16x4_f64 and the barrier
// and the LDS broadcast happen once per four pivots instead of once per pivot.
//
// Status: selectable with fedd_set_option("inv_kind", 1), NOT the default.  Measured on cfg 2
// (39 304 subdomains of <= 101 dofs): 6.8 ms against 5.3 ms for the scalar-pivot kernel.  MI355X's
// f64 matrix rate equals its f64 vector rate, so the rank-4 update costs the same issue cycles as
// four rank-1 updates (14 MFMAs of 64 cycles per step and wave against 4 x 49 FMAs of 4 cycles),
// the redundant 4 x 4 pivot-block inverse comes on top, and 172 VGPRs leave two workgroups per CU
// instead of three.  Kept as the checked alternative (tests/test_gpu_edge_cases.py).
//
// One workgroup (4 waves) per subdomain.  B = A^T is held in registers as 16 x 16 tiles in the MFMA
// accumulator layout (lane l, register q = element (row (l >> 4) + 4 q, column l & 15)); wave w owns
// the tile rows ti = w, w + 4, so one A-fragment serves a whole tile row and the T B-fragments are
// shared by the wave's tile rows.  Step s (pivots K = 4s .. 4s+3):
//   publish   rows K of B (register ks = s & 3 of tile row s >> 2) and columns K of B to LDS
//   barrier
//   every lane inverts the 4 x 4 pivot block D in registers (in-place Gauss-Jordan)
//   A-fragment(ti)[i][k] = +Dinv[i-4s][k] for i in K, else -(C[i][:] Dinv[:][k])
//   B-fragment(tj)[k][j] = I[k][j-4s]     for j in K, else  R[k][j]
//   rows K and columns K of B are cleared, then  B += A-fragment * B-fragment  for every tile,
// which leaves Dinv in B[K,K], Dinv R in B[K,:], -C Dinv in B[:,K] and B - C Dinv R elsewhere:
// one block step of the in-place Gauss-Jordan sweep.  Working on A^T makes the rows of A^-1 that
// the restricted operator needs come out as columns of B, so the slab [column c][row i] is written
// with 16 consecutive i per lane group (coalesced).
#include "fedd_internal.hpp"
#include <algorithm>

namespace fedd {
namespace {

constexpr int NMAX = SCHWARZ_NMAX;
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int bsearch_i32(const int32_t* a, int n, int32_t v) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int32_t m = a[mid];
        if (m == v) return mid;
        if (m < v) lo = mid + 1;
        else hi = mid - 1;
    }
    return -1;
}

__device__ __forceinline__ double pick4(const double4_t& v, int k) {
    return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
}

__device__ __forceinline__ void clear4(double4_t& v, int k) {
    v.x = k == 0 ? 0.0 : v.x;
    v.y = k == 1 ? 0.0 : v.y;
    v.z = k == 2 ? 0.0 : v.z;
    v.w = k == 3 ? 0.0 : v.w;
}

__device__ __forceinline__ double recip(double p) {
    double r = __builtin_amdgcn_rcp(p);  // about 23 bits, two Newton steps
    r = fma(fma(-p, r, 1.0), r, r);
    r = fma(fma(-p, r, 1.0), r, r);
    return r;
}

template <int T>
__global__ __launch_bounds__(256, 2) void k_invert_mfma(const int32_t* __restrict__ sub_n,
                                                        const int32_t* __restrict__ sub_nown,
                                                        const int32_t* __restrict__ sub_dofs,
                                                        const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ colind,
                                                        const double* __restrict__ val, int32_t n_rows,
                                                        int restricted, const int64_t* __restrict__ inv_ptr,
                                                        double* __restrict__ inv, int32_t* __restrict__ bad,
                                                        int n_lo, int n_hi) {
    constexpr int NP = 16 * T, TL = (T + 3) / 4;
    __shared__ int32_t sdof[NP];
    __shared__ double stage[16][NP + 1];
    __shared__ double Rraw[2][4][NP];
    __shared__ double Craw[2][NP][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = sub_n[b];
    if (n <= n_lo || n > n_hi) return;  // another size class handles this subdomain
    const int no = sub_nown[b];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, lr = lane >> 4, lc = lane & 15;
    for (int k = tid; k < NP; k += 256) sdof[k] = k < n ? sub_dofs[(int64_t)b * NMAX + k] : -1;
    __syncthreads();
    double4_t acc[TL][T];
    // ---- dense extraction of B = A^T: 16 rows of A (= 16 columns of B) at a time through LDS ----
    int32_t pb[T], pe[T], col0[T];
    double v0[T];
    {
        const int r = tid >> 4, l = tid & 15;
#pragma unroll
        for (int a = 0; a < T; ++a) {
            const int i = r + 16 * a;
            const int32_t g = i < n ? sdof[i] : -1;
            const bool stored = g >= 0 && g < n_rows;
            pb[a] = stored ? rowptr[g] + l : 0;
            pe[a] = stored ? rowptr[g + 1] : 0;
        }
#pragma unroll
        for (int a = 0; a < T; ++a) {
            const bool have = pb[a] < pe[a];
            col0[a] = have ? colind[pb[a]] : -1;
            v0[a] = have ? val[pb[a]] : 0.0;
        }
    }
#pragma unroll
    for (int a = 0; a < T; ++a) {
        for (int e = tid; e < 16 * (NP + 1); e += 256) (&stage[0][0])[e] = 0.0;
        __syncthreads();
        {
            const int r = tid >> 4, l = tid & 15;
            const int i = r + 16 * a;
            if (i < n) {
                const int32_t g = sdof[i];
                if (g < n_rows) {
                    int32_t col = col0[a];
                    double v = v0[a];
                    for (int32_t p = pb[a]; p < pe[a]; p += 16) {
                        if (p != pb[a]) {
                            col = colind[p];
                            v = val[p];
                        }
                        int cidx = bsearch_i32(sdof, no, col);
                        if (cidx < 0) {
                            cidx = bsearch_i32(sdof + no, n - no, col);
                            if (cidx >= 0) cidx += no;
                        }
                        if (cidx >= 0) stage[r][cidx] = v;
                    }
                } else if (l == 0) {
                    stage[r][i] = 1.0;  // ghost row (not stored on this rank): identity
                }
            } else if (l == 0) {
                stage[r][i] = 1.0;  // padding
            }
        }
        __syncthreads();
        // B[row][16 a + lc] = A[16 a + lc][row] = stage[lc][row]
#pragma unroll
        for (int tl = 0; tl < TL; ++tl) {
            const int ti = wave + 4 * tl;
            if (ti < T) {
                acc[tl][a].x = stage[lc][16 * ti + lr];
                acc[tl][a].y = stage[lc][16 * ti + lr + 4];
                acc[tl][a].z = stage[lc][16 * ti + lr + 8];
                acc[tl][a].w = stage[lc][16 * ti + lr + 12];
            } else {
                acc[tl][a] = double4_t{0.0, 0.0, 0.0, 0.0};
            }
        }
        __syncthreads();
    }
    // ---- block Gauss-Jordan sweep, four pivots per step ----
    bool singular = false;
    const int nsteps = (n + 3) >> 2;
    for (int s = 0; s < nsteps; ++s) {
        const int kt = s >> 2, ks = s & 3, buf = s & 1;
        // publish rows K and columns K of B
#pragma unroll
        for (int tl = 0; tl < TL; ++tl) {
            const int ti = wave + 4 * tl;
            if (ti >= T) continue;
            if (ti == kt) {
#pragma unroll
                for (int tj = 0; tj < T; ++tj) Rraw[buf][lr][16 * tj + lc] = pick4(acc[tl][tj], ks);
            }
#pragma unroll
            for (int tj = 0; tj < T; ++tj) {
                if (tj != kt) continue;  // uniform
                if ((lc >> 2) == ks) {
                    const int i = 16 * ti + lr;
                    Craw[buf][i][lc & 3] = acc[tl][tj].x;
                    Craw[buf][i + 4][lc & 3] = acc[tl][tj].y;
                    Craw[buf][i + 8][lc & 3] = acc[tl][tj].z;
                    Craw[buf][i + 12][lc & 3] = acc[tl][tj].w;
                }
            }
        }
        __syncthreads();
        // 4 x 4 pivot block, inverted in place (every lane, same values)
        double d[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) d[m][q] = Rraw[buf][m][4 * s + q];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            singular = singular || !(fabs(d[p][p]) > 1e-300);
            const double pinv = recip(d[p][p]);
#pragma unroll
            for (int q = 0; q < 4; ++q) d[p][q] = q == p ? pinv : d[p][q] * pinv;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m == p) continue;
                const double f = d[m][p];
#pragma unroll
                for (int q = 0; q < 4; ++q) d[m][q] = q == p ? -f * pinv : fma(-f, d[p][q], d[m][q]);
            }
        }
        // column lr of Dinv (this lane's k index of both fragments)
        double dk[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) dk[m] = lr == 0 ? d[m][0] : (lr == 1 ? d[m][1] : (lr == 2 ? d[m][2] : d[m][3]));
        // fragments
        double af[TL], bf[T];
#pragma unroll
        for (int tl = 0; tl < TL; ++tl) {
            const int ti = wave + 4 * tl;
            af[tl] = 0.0;
            if (ti >= T) continue;
            const int i = 16 * ti + lc;
            const double c0 = Craw[buf][i][0], c1 = Craw[buf][i][1], c2 = Craw[buf][i][2], c3 = Craw[buf][i][3];
            const double prod = -(fma(c3, dk[3], fma(c2, dk[2], fma(c1, dk[1], c0 * dk[0]))));
            const int m = lc & 3;
            const double own = m == 0 ? dk[0] : (m == 1 ? dk[1] : (m == 2 ? dk[2] : dk[3]));
            af[tl] = (i >> 2) == s ? own : prod;
        }
#pragma unroll
        for (int tj = 0; tj < T; ++tj) {
            const int j = 16 * tj + lc;
            const double rv = Rraw[buf][lr][j];
            bf[tj] = (j >> 2) == s ? ((j & 3) == lr ? 1.0 : 0.0) : rv;
        }
        // clear rows K and columns K, then the rank-4 update
#pragma unroll
        for (int tl = 0; tl < TL; ++tl) {
            const int ti = wave + 4 * tl;
            if (ti >= T) continue;
            if (ti == kt) {
#pragma unroll
                for (int tj = 0; tj < T; ++tj) clear4(acc[tl][tj], ks);
            }
#pragma unroll
            for (int tj = 0; tj < T; ++tj) {
                if (tj == kt) {  // uniform: one tile column per step
                    if ((lc >> 2) == ks) acc[tl][tj] = double4_t{0.0, 0.0, 0.0, 0.0};
                }
                acc[tl][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[tl], bf[tj], acc[tl][tj], 0, 0, 0);
            }
        }
        // no second barrier: the next step writes the other LDS buffer, and the buffer written two
        // steps from now was last read before the barrier of the step in between
    }
    if (singular && tid == 0) bad[0] = 1;
    // ---- needed rows of A^-1 = columns of B -> slab [column c of A^-1][row i], i fastest ----
    const int nrow = restricted ? no : n;
    double* __restrict__ slab = inv + inv_ptr[b];
#pragma unroll
    for (int tl = 0; tl < TL; ++tl) {
        const int ti = wave + 4 * tl;
        if (ti >= T) continue;
#pragma unroll
        for (int tj = 0; tj < T; ++tj) {
            const int i = 16 * tj + lc;  // row of A^-1
            if (i >= nrow) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 16 * ti + lr + 4 * q;  // column of A^-1
                if (c < n) slab[(int64_t)c * nrow + i] = pick4(acc[tl][tj], q);
            }
        }
    }
}

}  // namespace

// size classes n <= 16 T, T = 2, 4, 6, 7, 8; subdomains above 128 dofs are left to the caller
int schwarz_invert_mfma(fedd_ctx* c, int restricted, int32_t* d_bad, int max_n) {
    const dim3 grid((unsigned)c->sw_nsub), blk(256);
    const int32_t n_rows = (int32_t)c->n_rows_ext;   // rows the local matrices can read (owned + row ghosts)
#define INV_MFMA(T, LO, HI)                                                                                    \
    if (max_n > (LO))                                                                                          \
        hipLaunchKernelGGL(k_invert_mfma<T>, grid, blk, 0, c->stream, (const int32_t*)c->d_sub_n.p,            \
                           (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p,                   \
                           (const int32_t*)c->d_rowptr.p, (const int32_t*)c->d_colind.p,                       \
                           (const double*)c->d_val.p, n_rows, restricted, (const int64_t*)c->d_inv_ptr.p,      \
                           c->d_inv.p, d_bad, (LO), (HI))
    INV_MFMA(2, 0, 32);
    INV_MFMA(4, 32, 64);
    INV_MFMA(6, 64, 96);
    INV_MFMA(7, 96, 112);
    INV_MFMA(8, 112, 128);
#undef INV_MFMA
    FEDD_HIP(hipGetLastError());
    return 0;
}

}  // namespace fedd
