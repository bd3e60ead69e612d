// Right-preconditioned restarted GMRES on the device, block size 1.
//
// Stands in for Thyra::solve on the Belos "Block GMRES" LOWS that
// LinearSolver::solveMonolithic builds (feddlib/problems/Solver/LinearSolver_def.hpp:72-135) with
// the settings of feddlib/problems/tests/laplace/parametersSolver.xml:5-15 (Block Size 1, DGKS
// orthogonalisation, relative residual tolerance, maximum iterations) and the preconditioner
// attached as Thyra "unspecified" side = right.  Belos is not in the reference tree; this is the
// published algorithm (Saad, Iterative Methods, Alg. 9.5):
//   - classical Gram-Schmidt with two passes (what Belos' DGKS / ICGS managers do), default
//     formulation: the second pass of a basis vector is delayed and fused with the first pass of
//     the next Krylov vector (DCGS2, see k_multidot2 below): two sweeps over the basis and one
//     reduction per iteration.  fedd_set_option("gmres_kind", 1) selects the plain two-pass form
//     (fused multi-dot + multi-axpy twice, second pass gated by the DGKS test on the device);
//   - Givens QR of the Hessenberg matrix on the device (single lane), the host reads one double
//     per iteration (implicit relative residual) to decide convergence, one iteration behind;
//   - x = x0 + M^-1 (V y) at the end of a cycle (M is a fixed linear operator, so Z is not stored).
// All sums run in a fixed order: results are bitwise reproducible run to run.
#include "fedd_internal.hpp"
#include <rccl/rccl.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace fedd {
namespace {

constexpr int MD_ROWS = 2048;  // rows per workgroup of the multi-dot (256 lanes x 4 x double2)
constexpr int MD_CG = 8;       // basis columns per workgroup of the multi-dot
constexpr int MD2_CG = 8;      // basis columns per workgroup of the two-vector multi-dot (16: 13 % slower)
constexpr int AX_ROWS = 512;   // rows per workgroup of the multi-axpy (256 lanes x double2)

// cached load for the work vectors (u, w): every column group of the multi-dot re-reads them
__device__ __forceinline__ double2 ld2c(const double* __restrict__ p, int64_t r, int64_t n) {
    if (r + 1 < n) return *reinterpret_cast<const double2*>(p + r);
    double2 v;
    v.x = r < n ? p[r] : 0.0;
    v.y = 0.0;
    return v;
}

__device__ __forceinline__ double2 ld2(const double* __restrict__ p, int64_t r, int64_t n) {
    // basis columns are streamed (each is read twice per iteration, 0.8 GB apart): non-temporal loads
    // keep them from displacing the vectors and the matrix in L2 / Infinity Cache (-1.5 ms per step)
    if (r + 1 < n) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + r));
        double2 v;
        v.x = t.x;
        v.y = t.y;
        return v;
    }
    double2 v;
    v.x = r < n ? p[r] : 0.0;
    v.y = 0.0;
    return v;
}

// partial[col*nblk + blk] = sum over the workgroup's rows of V_col . w ; col == ncolsV means w . w.
// A workgroup keeps its 2048 rows of w in registers and sweeps MD_CG basis columns, so the basis
// is read from HBM exactly once per pass with 16-byte loads; w is re-read from L2.
__global__ __launch_bounds__(256) void k_multidot(const double* __restrict__ V, int64_t ldv, int64_t n, int ncolsV,
                                                  const double* __restrict__ w, double* __restrict__ partial,
                                                  int nblk, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    __shared__ double sh[4][MD_CG];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * MD_ROWS + 2 * tid;
    double2 wv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[k] = ld2c(w, r0 + 512 * k, n);
    double acc[MD_CG];
#pragma unroll
    for (int cc = 0; cc < MD_CG; ++cc) {
        const int col = blockIdx.y * MD_CG + cc;
        double s = 0.0;
        if (col <= ncolsV) {
            const double* __restrict__ a = col < ncolsV ? V + (int64_t)col * ldv : w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double2 v = ld2(a, r0 + 512 * k, n);
                s += v.x * wv[k].x + v.y * wv[k].y;
            }
        }
        acc[cc] = s;
    }
#pragma unroll
    for (int cc = 0; cc < MD_CG; ++cc) {
        double s = acc[cc];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((tid & 63) == 0) sh[tid >> 6][cc] = s;
    }
    __syncthreads();
    if (tid < MD_CG) {
        const int col = blockIdx.y * MD_CG + tid;
        if (col <= ncolsV) partial[(int64_t)col * nblk + blockIdx.x] = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
    }
}

// out[col] = sum_blk partial[col][blk]   (one workgroup per column, fixed order)
__global__ __launch_bounds__(256) void k_reduce_cols(const double* __restrict__ partial, double* __restrict__ out,
                                                     int nblk, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    __shared__ double sh[256];
    const int col = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256) s += partial[(int64_t)col * nblk + k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[col] = sh[0];
}

// w -= sum_c h[c] V_c ; also partial sums of ||w_new||^2 into partial[blk].  One double2 of w per
// lane, the column loop unrolled 8x keeps 8 x 16 B loads in flight per lane.
__global__ __launch_bounds__(256) void k_multiaxpy(const double* __restrict__ V, int64_t ldv, int64_t n, int ncols,
                                                   const double* __restrict__ h, double* __restrict__ w,
                                                   double* __restrict__ partial, const int32_t* __restrict__ gate) {
    if (gate && !*gate) return;
    __shared__ double sh_h[1024];
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    for (int c = tid; c < ncols; c += 256) sh_h[c] = h[c];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * AX_ROWS + 2 * tid;
    double2 v = ld2c(w, r, n);
    int c = 0;
    for (; c + 8 <= ncols; c += 8) {
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ld2(V + (int64_t)(c + u) * ldv, r, n);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v.x -= sh_h[c + u] * t[u].x;
            v.y -= sh_h[c + u] * t[u].y;
        }
    }
    for (; c < ncols; ++c) {
        const double2 t = ld2(V + (int64_t)c * ldv, r, n);
        v.x -= sh_h[c] * t.x;
        v.y -= sh_h[c] * t.y;
    }
    double nrm = 0.0;
    if (r + 1 < n) {
        *reinterpret_cast<double2*>(w + r) = v;
        nrm = v.x * v.x + v.y * v.y;
    } else if (r < n) {
        w[r] = v.x;
        nrm = v.x * v.x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
    if ((tid & 63) == 0) sh[tid >> 6] = nrm;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// scalars layout in d_small (doubles): see Offsets below
struct Off {
    int H, cs, sn, g, h1, h2, nrm, y, misc;
};

// DGKS test from the pass-1 coefficients alone: with an orthonormal basis ||w - V h||^2 =
// ||w||^2 - ||h||^2, so the second pass is needed iff ||h||^2 > 1/2 ||w||^2 (h1[0..j] = V^T w,
// h1[j+1] = w.w, all already reduced over ranks).  When the test does not fire the difference has
// no cancellation and is the squared norm after the pass (nrm1); when it fires the norm comes from
// the second pass.  No extra reduction, no extra all-reduce.
__global__ void k_dgks_gate(const double* __restrict__ h1, int j, int32_t* __restrict__ gate, double* __restrict__ nrm1) {
    double s = 0.0;
    for (int c = 0; c <= j; ++c) s += h1[c] * h1[c];
    const double ww = h1[j + 1];
    gate[0] = s > 0.5 * ww ? 1 : 0;
    nrm1[0] = fmax(ww - s, 0.0);
}

// Finish column j of the Hessenberg matrix: h = h1 (+ h2), h_{j+1,j} = ||w||; apply the previous
// rotations, create the new one, update g.  misc[0] = |g_{j+1}|, misc[1] = 1/h_{j+1,j}.
// One workgroup: all lanes stage the column and the stored rotations in LDS (independent loads),
// lane 0 then runs the inherently sequential rotation chain out of LDS instead of a chain of
// dependent global loads.
__global__ __launch_bounds__(256) void k_givens(double* __restrict__ S, Off o, int j, int m,
                                                const int32_t* __restrict__ gate,
                                                const double* __restrict__ nrm2_partial, int npart) {
    extern __shared__ double gs[];  // hcol[m+2] | cs[m] | sn[m]
    __shared__ double red[256];
    double* hc = gs;
    double* lcs = hc + (m + 2);
    double* lsn = lcs + m;
    const bool two = gate[0] != 0;
    // one rank: the partial sums of ||w||^2 of the second pass are added here (fixed order) instead
    // of in a launch of their own; several ranks reduce and all-reduce before this kernel
    if (nrm2_partial && two) {
        double s = 0.0;
        for (int k = threadIdx.x; k < npart; k += 256) s += nrm2_partial[k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
            __syncthreads();
        }
        if (threadIdx.x == 0) S[o.nrm + 2] = red[0];
    }
    for (int i = threadIdx.x; i <= j; i += blockDim.x) hc[i] = S[o.h1 + i] + (two ? S[o.h2 + i] : 0.0);
    for (int i = threadIdx.x; i < j; i += blockDim.x) {
        lcs[i] = S[o.cs + i];
        lsn[i] = S[o.sn + i];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double hn = sqrt(two ? S[o.nrm + 2] : S[o.nrm + 1]);
    hc[j + 1] = hn;
    for (int i = 0; i < j; ++i) {
        const double t = lcs[i] * hc[i] + lsn[i] * hc[i + 1];
        hc[i + 1] = -lsn[i] * hc[i] + lcs[i] * hc[i + 1];
        hc[i] = t;
    }
    const double d = hypot(hc[j], hc[j + 1]);
    const double cj = d > 0 ? hc[j] / d : 1.0, sj = d > 0 ? hc[j + 1] / d : 0.0;
    S[o.cs + j] = cj;
    S[o.sn + j] = sj;
    hc[j] = d;
    double* H = S + o.H + (int64_t)j * (m + 1);
    for (int i = 0; i <= j; ++i) H[i] = hc[i];
    H[j + 1] = 0.0;
    const double gj = S[o.g + j];
    S[o.g + j + 1] = -sj * gj;
    S[o.g + j] = cj * gj;
    S[o.misc + 0] = fabs(sj * gj);
    S[o.misc + 1] = hn > 0 ? 1.0 / hn : 0.0;
    S[o.misc + 2] = hn;
}

__global__ void k_scale_to(const double* __restrict__ w, const double* __restrict__ scal, double* __restrict__ out,
                           int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = w[i] * scal[0];
}

// start of a cycle: beta = sqrt(nrm2[0]); V0 = r / beta; g = (beta, 0, ...)
__global__ void k_cycle_init(double* __restrict__ S, Off o, int m, const double* __restrict__ rr) {
    const double beta = sqrt(rr[0]);
    for (int i = 0; i <= m; ++i) S[o.g + i] = 0.0;
    S[o.g] = beta;
    S[o.misc + 1] = beta > 0 ? 1.0 / beta : 0.0;
    S[o.misc + 3] = beta;
}

// y = R^-1 g for the k x k upper triangle (column-major H with leading dimension m+1)
// Column-oriented back substitution by one workgroup: y_i = t_i / R_ii, then every lane r < i
// updates t_r -= R_ri y_i (column i of H is contiguous).  Fixed order => reproducible.
__global__ __launch_bounds__(256) void k_backsolve(double* __restrict__ S, Off o, int k, int m) {
    extern __shared__ double t[];  // [k]
    const double* H = S + o.H;
    for (int i = threadIdx.x; i < k; i += blockDim.x) t[i] = S[o.g + i];
    __syncthreads();
    for (int i = k - 1; i >= 0; --i) {
        const double yi = t[i] / H[(int64_t)i * (m + 1) + i];
        __syncthreads();
        if (threadIdx.x == 0) S[o.y + i] = yi;
        for (int r = threadIdx.x; r < i; r += blockDim.x) t[r] -= H[(int64_t)i * (m + 1) + r] * yi;
        __syncthreads();
    }
}

// u = sum_c y[c] V_c : one double2 of u per lane, the column loop unrolled 8x (8 x 16 B non-temporal loads in flight per lane;
// the first version, a scalar load per lane and column, ran at half the bandwidth: 1.1 ms for 45 columns at 9.9 M rows)
__global__ __launch_bounds__(256) void k_combine(const double* __restrict__ V, int64_t ldv, int64_t n, int k,
                                                 const double* __restrict__ y, double* __restrict__ u) {
    __shared__ double sh_y[1024];
    const int tid = threadIdx.x;
    for (int c = tid; c < k; c += 256) sh_y[c] = y[c];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * AX_ROWS + 2 * tid;
    double2 v = {0.0, 0.0};
    int c = 0;
    for (; c + 8 <= k; c += 8) {
        double2 t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = ld2(V + (int64_t)(c + q) * ldv, r, n);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v.x += sh_y[c + q] * t[q].x;
            v.y += sh_y[c + q] * t[q].y;
        }
    }
    for (; c < k; ++c) {
        const double2 t = ld2(V + (int64_t)c * ldv, r, n);
        v.x += sh_y[c] * t.x;
        v.y += sh_y[c] * t.y;
    }
    if (r + 1 < n) *reinterpret_cast<double2*>(u + r) = v;
    else if (r < n) u[r] = v.x;
}

// ------------------------------------------------------------------------------------------------
// Delayed re-orthogonalisation (DCGS2; Bielich, Langou, Thomas, Swirydowicz, Yamazaki, Boman,
// "Low-synch Gram-Schmidt with delayed reorthogonalization for Krylov solvers", 2022): the second
// Gram-Schmidt pass of basis vector k+1 and the first pass of the next Krylov vector are one sweep
// of dot products and one sweep of updates over the basis, instead of two and two.
//
// State before a step: V_k = [v_1 .. v_k] final (orthonormal), u = first-pass result that becomes
// v_{k+1}, hp = first-pass coefficients of column k of H.  With B = A M^-1:
//   wt = B u                                        (u is NOT yet re-orthogonalised or normalised)
//   sweep 1:  s = V_k^T u,  t = V_k^T wt,  alpha2 = u.u,  gamma = u.wt      (one reduction)
//   beta = sqrt(alpha2 - s.s);  column k of H = [hp + s ; beta]   -> Givens, residual
//   v_{k+1} = (u - V_k s) / beta,   B v_{k+1} = (wt - V_{k+1} Hbar_k s) / beta   (B V_k = V_{k+1} Hbar_k)
//   hp'_i = (t_i - (Hbar_k s)_i) / beta  (i <= k),  hp'_{k+1} = ((gamma - s.t)/beta - (Hbar_k s)_{k+1}) / beta
//   sweep 2:  v_{k+1} as above,  u' = (wt - V_k t)/beta - v_{k+1} (gamma - s.t)/beta^2
// u' is the first-pass result for the next vector; every basis column is read twice per iteration.

// partial[(2 col + which) * nblk + blk]: which = 0: V_col . u, 1: V_col . w; col == ncolsV: u.u and u.w
template <int NCH>
__global__ __launch_bounds__(256) void k_multidot2(const double* __restrict__ V, int64_t ldv, int64_t n, int ncolsV,
                                                   const double* __restrict__ u, const double* __restrict__ w,
                                                   double* __restrict__ partial, int nblk) {
    __shared__ double sh[4][2 * MD2_CG];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * (512 * NCH) + 2 * tid;
    double2 uv[NCH], wv[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        uv[k] = ld2c(u, r0 + 512 * k, n);
        wv[k] = ld2c(w, r0 + 512 * k, n);
    }
    // column groups cg = blockIdx.y, blockIdx.y + gridDim.y, ...: u and w stay in registers across them
    for (int cg = blockIdx.y; cg * MD2_CG <= ncolsV; cg += gridDim.y) {
        double au[MD2_CG], aw[MD2_CG];
#pragma unroll
        for (int cc = 0; cc < MD2_CG; ++cc) {
            const int col = cg * MD2_CG + cc;
            double su = 0.0, sw = 0.0;
            if (col <= ncolsV) {
                const double* __restrict__ a = col < ncolsV ? V + (int64_t)col * ldv : u;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const double2 v = ld2(a, r0 + 512 * k, n);
                    su += v.x * uv[k].x + v.y * uv[k].y;
                    sw += v.x * wv[k].x + v.y * wv[k].y;
                }
            }
            au[cc] = su;
            aw[cc] = sw;
        }
#pragma unroll
        for (int cc = 0; cc < MD2_CG; ++cc) {
            double su = au[cc], sw = aw[cc];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                su += __shfl_down(su, off, 64);
                sw += __shfl_down(sw, off, 64);
            }
            if ((tid & 63) == 0) {
                sh[tid >> 6][2 * cc] = su;
                sh[tid >> 6][2 * cc + 1] = sw;
            }
        }
        __syncthreads();
        if (tid < 2 * MD2_CG) {
            const int col = cg * MD2_CG + (tid >> 1);
            if (col <= ncolsV)
                partial[(int64_t)(2 * col + (tid & 1)) * nblk + blockIdx.x] = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
        }
        __syncthreads();
    }
}

struct Off2 {
    int Hraw, hp, st, cf;  // unrotated Hessenberg (m+1) x m, first-pass column, reduced dots, sweep-2 coefficients
};

// The small algebra of a DCGS2 step (one workgroup): finalise column kc = k - 1 of H, rotate it,
// update g and the residual, and prepare the coefficients of sweep 2 and the next first-pass column.
// misc[0] = |g_k| (residual), misc[2] = beta (0: breakdown), cf = [s (m) | t (m) | 1/beta | c_last].
// host_out (pinned host memory, device-visible): the three numbers the host's lagged convergence test reads -- written by
// the kernel itself, so that no copy engine call sits in the stream between two iterations (8.7 us each)
__global__ __launch_bounds__(256) void k_dcgs2_small(double* __restrict__ S, Off o, Off2 o2, int k, int m,
                                                     double* __restrict__ host_out) {
    extern __shared__ double sm[];  // s[m] | t[m] | hc[m+2] | lcs[m] | lsn[m] | Hs[m+1]
    __shared__ double red[2][256];
    double* s = sm;
    double* t = s + m;
    double* hc = t + m;
    double* lcs = hc + (m + 2);
    double* lsn = lcs + m;
    double* Hs = lsn + m;
    const int tid = threadIdx.x, kc = k - 1, ldh = m + 1;
    const double* st = S + o2.st;
    double ps = 0.0, pd = 0.0;
    for (int i = tid; i < k; i += 256) {
        const double si = st[2 * i], ti = st[2 * i + 1];
        s[i] = si;
        t[i] = ti;
        ps += si * si;
        pd += si * ti;
    }
    for (int i = tid; i < kc; i += 256) {
        lcs[i] = S[o.cs + i];
        lsn[i] = S[o.sn + i];
    }
    red[0][tid] = ps;
    red[1][tid] = pd;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if (tid < q) {
            red[0][tid] += red[0][tid + q];
            red[1][tid] += red[1][tid + q];
        }
        __syncthreads();
    }
    const double ss = red[0][0], sd = red[1][0];
    const double alpha2 = st[2 * k], gamma = st[2 * k + 1];
    const double beta2 = alpha2 - ss;
    const double beta = beta2 > 0.0 ? sqrt(beta2) : 0.0;
    const double ib = beta > 0.0 ? 1.0 / beta : 0.0;
    // unrotated column kc
    double* Hraw = S + o2.Hraw;
    for (int i = tid; i <= k; i += 256) {
        const double v = i < k ? S[o2.hp + i] + s[i] : beta;
        Hraw[(int64_t)kc * ldh + i] = v;
        hc[i] = v;
    }
    __syncthreads();
    // Hs = Hbar_k s  (rows 0..k, columns 0..kc; entries below the sub-diagonal are zero)
    for (int i = tid; i <= k; i += 256) {
        // (eight independent loads in flight: the Hessenberg matrix is read from L2, and a chain of
        // dependent-latency loads was the whole cost of this kernel)
        double a = 0.0;
        int c = i > 0 ? i - 1 : 0;
        for (; c + 8 <= kc; c += 8) {
            double hv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) hv[q] = Hraw[(int64_t)(c + q) * ldh + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) a += hv[q] * s[c + q];
        }
        for (; c < kc; ++c) a += Hraw[(int64_t)c * ldh + i] * s[c];
        a += hc[i] * s[kc];  // the column written above, taken from LDS
        Hs[i] = a;
    }
    __syncthreads();
    // next first-pass column and the coefficients of sweep 2
    for (int i = tid; i <= k; i += 256) {
        const double v = i < k ? (t[i] - Hs[i]) * ib : ((gamma - sd) * ib - Hs[k]) * ib;
        S[o2.hp + i] = v;
    }
    for (int i = tid; i < k; i += 256) {
        S[o2.cf + i] = s[i];
        S[o2.cf + m + i] = t[i];
    }
    if (tid == 0) {
        S[o2.cf + 2 * m] = ib;
        S[o2.cf + 2 * m + 1] = (gamma - sd) * ib * ib;
        // Givens: previous rotations on the new column, new rotation, g and the residual
        for (int i = 0; i < kc; ++i) {
            const double a = lcs[i] * hc[i] + lsn[i] * hc[i + 1];
            hc[i + 1] = -lsn[i] * hc[i] + lcs[i] * hc[i + 1];
            hc[i] = a;
        }
        const double d = hypot(hc[kc], hc[kc + 1]);
        const double cj = d > 0 ? hc[kc] / d : 1.0, sj = d > 0 ? hc[kc + 1] / d : 0.0;
        S[o.cs + kc] = cj;
        S[o.sn + kc] = sj;
        hc[kc] = d;
        double* H = S + o.H + (int64_t)kc * ldh;
        for (int i = 0; i <= kc; ++i) H[i] = hc[i];
        H[kc + 1] = 0.0;
        const double gj = S[o.g + kc];
        S[o.g + kc + 1] = -sj * gj;
        S[o.g + kc] = cj * gj;
        S[o.misc + 0] = fabs(sj * gj);
        S[o.misc + 1] = ib;
        S[o.misc + 2] = beta;
        if (host_out) {
            host_out[0] = fabs(sj * gj);
            host_out[1] = ib;
            host_out[2] = beta;
            __threadfence_system();
        }
    }
}

// sweep 2: V[:, k] = (u - V_k s) / beta ;  u <- (w - V_k t) / beta - V[:, k] * c_last
__global__ __launch_bounds__(256) void k_axpy2(double* __restrict__ V, int64_t ldv, int64_t n, int k,
                                               const double* __restrict__ cf, int m, double* __restrict__ u,
                                               const double* __restrict__ w) {
    __shared__ double sh_s[1024], sh_t[1024];
    const int tid = threadIdx.x;
    for (int c = tid; c < k; c += 256) {
        sh_s[c] = cf[c];
        sh_t[c] = cf[m + c];
    }
    __syncthreads();
    const double ib = cf[2 * m], cl = cf[2 * m + 1];
    const int64_t r = (int64_t)blockIdx.x * AX_ROWS + 2 * tid;
    double2 as = {0.0, 0.0}, at = {0.0, 0.0};
    int c = 0;
    for (; c + 8 <= k; c += 8) {
        double2 q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = ld2(V + (int64_t)(c + j) * ldv, r, n);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            as.x += sh_s[c + j] * q[j].x;
            as.y += sh_s[c + j] * q[j].y;
            at.x += sh_t[c + j] * q[j].x;
            at.y += sh_t[c + j] * q[j].y;
        }
    }
    for (; c < k; ++c) {
        const double2 q = ld2(V + (int64_t)c * ldv, r, n);
        as.x += sh_s[c] * q.x;
        as.y += sh_s[c] * q.y;
        at.x += sh_t[c] * q.x;
        at.y += sh_t[c] * q.y;
    }
    const double2 uu = ld2c(u, r, n), ww = ld2c(w, r, n);
    double2 vn, un;
    vn.x = (uu.x - as.x) * ib;
    vn.y = (uu.y - as.y) * ib;
    un.x = (ww.x - at.x) * ib - vn.x * cl;
    un.y = (ww.y - at.y) * ib - vn.y * cl;
    double* vk = V + (int64_t)k * ldv;
    if (r + 1 < n) {
        {
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d tv;
            tv.x = vn.x;
            tv.y = vn.y;
            __builtin_nontemporal_store(tv, reinterpret_cast<v2d*>(vk + r));  // next read is a basis sweep
        }
        *reinterpret_cast<double2*>(u + r) = un;
    } else if (r < n) {
        vk[r] = vn.x;
        u[r] = un.x;
    }
}

// out = m o a + (1 - m) o b  (m = 0 / 1 mask): the constrained operator of the interior solves, see gmres_solve
__global__ void k_mask_mix(const double* __restrict__ m, const double* __restrict__ a, const double* __restrict__ b,
                           double* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = m[i] != 0.0 ? a[i] : b[i];
}

__global__ void k_axpby(double a, const double* __restrict__ x, double b, const double* __restrict__ y,
                        double* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i];
}

}  // namespace

int allreduce_sum(fedd_ctx* c, double* d_buf, int n) {
    if (c->nranks == 1) return 0;
    ScopedTimer timer(c, FEDD_T_ALLREDUCE);
    if (c->cb_allreduce) {  // host-staged transport (functional tests)
        std::vector<double> tmp((size_t)n);
        FEDD_HIP(hipMemcpyAsync(tmp.data(), d_buf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        const int rc = c->cb_allreduce(c->cb_user, tmp.data(), n);
        FEDD_CHECK(rc == 0, "allreduce: the host callback failed (%d)", rc);
        FEDD_HIP(hipMemcpyAsync(d_buf, tmp.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        return 0;
    }
    FEDD_CHECK(c->comm, "allreduce: no communicator");
    ncclResult_t r = ncclAllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream);
    FEDD_CHECK(r == ncclSuccess, "ncclAllReduce: %s", ncclGetErrorString(r));
    return 0;
}

// GMRES with the delayed second Gram-Schmidt pass (kernels and formulas above).  Same iterates as
// the two-pass variant below in exact arithmetic; one operator application more per restart cycle
// (the lag), half the passes over the basis and one all-reduce per iteration.
static int gmres_solve_dcgs2(fedd_ctx* c, const double* d_b, double* d_x, double rtol, int max_it, int restart,
                             int use_prec, int* its_out, double* relres_out) {
    const int64_t n = c->n_rows;
    const int m = std::min(restart, max_it);
    const int64_t ldv = (n + 15) & ~(int64_t)15;
    FEDD_CHECK(m + 2 <= 1024, "gmres: restart length above 1022 is not supported");
    const int nblk = (int)((n + MD_ROWS - 1) / MD_ROWS), nblk2 = (int)((n + AX_ROWS - 1) / AX_ROWS);
    FEDD_TRY(c->d_V.ensure((size_t)(m + 1) * ldv));
    if (c->gm_V_ldv != ldv) c->gm_V_ldv = -1;   // another column layout: the s-step solver clears the basis before its next use
    // work vectors carry room for the ghost entries behind the owned ones (several ranks): the halo import of an
    // operator input then goes straight into the vector, without a copy into a column-length buffer first
    const int64_t nc = (std::max<int64_t>(n, c->n_cols) + 15) & ~(int64_t)15;
    FEDD_TRY(c->d_w.ensure(std::max<size_t>((size_t)2 * nc, c->d_w.cap)));   // u | w~
    FEDD_TRY(c->d_Z.ensure((size_t)nc * 2));
    FEDD_TRY(c->d_part.ensure(std::max((size_t)(2 * m + 4) * nblk * 2, (size_t)nblk2)));
    Off o;
    Off2 o2;
    int p = 0;
    o.H = p; p += (m + 1) * m;
    o.cs = p; p += m;
    o.sn = p; p += m;
    o.g = p; p += m + 1;
    o.h1 = p; p += m + 2;
    o.h2 = p; p += m + 2;
    o.nrm = p; p += 4;
    o.y = p; p += m;
    o.misc = p; p += 8;
    o2.Hraw = p; p += (m + 1) * m;
    o2.hp = p; p += m + 2;
    o2.st = p; p += 2 * m + 4;
    o2.cf = p; p += 2 * m + 4;
    FEDD_TRY(c->d_small.ensure((size_t)p + 8));
    double* S = c->d_small.p;
    double* V = c->d_V.p;
    double* u = c->d_w.p;        // first-pass result / next basis vector before its second pass
    double* wt = c->d_w.p + nc;  // B u
    double* z = c->d_Z.p;        // M^-1 v
    double* r = c->d_Z.p + nc;   // residual / V y
    const dim3 gn((unsigned)((n + 255) / 256)), blk(256);
    hipStream_t st = c->stream;

    auto norm2_into = [&](const double* v, double* out) -> int {
        hipLaunchKernelGGL(k_multidot, dim3(nblk, 1), blk, 0, st, v, ldv, n, 0, v, c->d_part.p, nblk, (const int32_t*)nullptr);
        hipLaunchKernelGGL(k_reduce_cols, dim3(1), blk, 0, st, (const double*)c->d_part.p, out, nblk, (const int32_t*)nullptr);
        return allreduce_sum(c, out, 1);
    };
    // c->gm_mask != nullptr: the constrained system  A^ = D A D + (I - D),  M^^-1 = D M^-1 D + (I - D)  with
    // D = diag(mask): dofs with mask 0 are held at the value the right-hand side gives them and decouple (the
    // discrete harmonic extensions of the GDSW coarse space: interface dofs held, interiors solved).  The inputs
    // already carry D x = x wherever it matters: in = D in + (I - D) in.
    const double* mk = c->gm_mask;
    auto apply_B = [&](const double* in, double* out) -> int {  // out = A M^-1 in
        const bool tail = in == u;     // u, z and r have a ghost tail, a basis column does not
        if (use_prec) FEDD_TRY(schwarz_apply(c, in, z, tail));
        if (mk) {
            // z <- D M^-1 D in + (I - D) in   (M^-1 D in: the masked entries of `in` are excluded by a copy)
            if (use_prec) hipLaunchKernelGGL(k_mask_mix, gn, blk, 0, st, mk, (const double*)z, in, z, n);
            else FEDD_HIP(hipMemcpyAsync(z, in, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
            FEDD_TRY(spmv_owned(c, z, out, true));
            hipLaunchKernelGGL(k_mask_mix, gn, blk, 0, st, mk, (const double*)out, (const double*)z, out, n);
            return 0;
        }
        return spmv_owned(c, use_prec ? z : in, out, use_prec ? true : tail);
    };

    if (c->gm_x0 && !mk) {   // "Zero Initial Guess" = false (LinearSolver_def.hpp:76-78): d_x holds x_0, r_0 = b - A x_0
        FEDD_TRY(spmv_owned(c, d_x, wt));
        hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, d_b, -1.0, (const double*)wt, r, n);
    } else {
        FEDD_HIP(hipMemsetAsync(d_x, 0, (size_t)n * sizeof(double), st));  // "Zero Initial Guess" (LinearSolver_def.hpp:76-78)
        FEDD_HIP(hipMemcpyAsync(r, d_b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    FEDD_TRY(norm2_into(r, S + o.nrm + 3));
    FEDD_HIP(hipMemcpyAsync(c->h_pinned, S + o.nrm + 3, sizeof(double), hipMemcpyDeviceToHost, st));
    FEDD_HIP(hipStreamSynchronize(st));
    const double beta0 = std::sqrt(c->h_pinned[0]);
    int its = 0;
    double relres = beta0 > 0 ? 1.0 : 0.0;
    if (!(beta0 > 0)) {
        if (its_out) *its_out = 0;
        if (relres_out) *relres_out = 0.0;
        return 0;
    }
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int q = 0; q < 2; ++q) FEDD_HIP(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));
    struct EvGuard {
        hipEvent_t* e;
        ~EvGuard() {
            for (int q = 0; q < 2; ++q)
                if (e[q]) (void)hipEventDestroy(e[q]);
        }
    } ev_guard{ev};
    // launch shape of k_multidot2: 2048 rows per workgroup, one grid row per group of 8 columns.
    // Measured alternatives on cfg 2 (ms per step): 1024 rows 43.9, column groups looped inside one
    // grid row 46.4 (2048 rows) / 44.5 (1024 rows), 2 or 4 grid rows 42.7-44.6; this one 43.1.
    // grid rows of k_multidot2 = column groups in flight per row block: every workgroup reads its rows of u and B u once
    // and keeps them across its column groups, so fewer grid rows = fewer re-reads of those two vectors (PMC: 1.24x the
    // algorithmic bytes with one group per workgroup at 214^3); enough of them to fill the GPU when the vectors are short
    const int md2_nch = c->md2_nch == 2 ? 2 : 4;
    const int64_t nblkd_ = (n + 512 * md2_nch - 1) / (512 * md2_nch);
    const int md2_gy = c->md2_gy > 0 ? c->md2_gy : (int)std::max<int64_t>(1, (2048 + nblkd_ - 1) / nblkd_);
    const int nblkd = (int)((n + 512 * md2_nch - 1) / (512 * md2_nch));
    bool converged = false;
    while (!converged && its < max_it) {
        // v_1 = r / ||r||, then the (not delayed) first pass of B v_1: hp = v_1 . w, u = w - v_1 hp
        hipLaunchKernelGGL(k_cycle_init, dim3(1), dim3(1), 0, st, S, o, m, (const double*)(S + o.nrm + 3));
        hipLaunchKernelGGL(k_scale_to, gn, blk, 0, st, (const double*)r, (const double*)(S + o.misc + 1), V, n);
        FEDD_TRY(apply_B(V, u));
        {
            ScopedTimer t(c, FEDD_T_ORTHO);
            hipLaunchKernelGGL(k_multidot, dim3(nblk, 1), blk, 0, st, (const double*)V, ldv, n, 1, (const double*)u,
                               c->d_part.p, nblk, (const int32_t*)nullptr);
            hipLaunchKernelGGL(k_reduce_cols, dim3(1), blk, 0, st, (const double*)c->d_part.p, S + o2.hp, nblk,
                               (const int32_t*)nullptr);
            FEDD_TRY(allreduce_sum(c, S + o2.hp, 1));
            hipLaunchKernelGGL(k_multiaxpy, dim3(nblk2), blk, 0, st, (const double*)V, ldv, n, 1, (const double*)(S + o2.hp),
                               u, c->d_part.p, (const int32_t*)nullptr);
            t.stop();
        }
        int kfin = 0;  // finalised columns of this cycle
        int issued = its, checked = 0, queued = 0;
        auto check = [&](int jj) -> int {
            if (hipEventSynchronize(ev[jj & 1]) != hipSuccess) return -1;
            const double* hp = c->h_pinned + 4 * (jj & 1);
            ++its;
            ++checked;
            kfin = jj + 1;
            relres = hp[0] / beta0;
            // beta^2 = u.u - s.s <= 0: a lucky breakdown, or cancellation after the first pass lost
            // orthogonality.  Only the implicit residual says "converged"; otherwise the cycle ends here and
            // the true residual decides (below).
            const bool breakdown = !(hp[2] > 0.0);
            return relres <= rtol ? 1 : (breakdown ? 2 : 0);
        };
        bool broke = false;
        for (int j = 0; j < m && issued < max_it; ++j) {
            const int k = j + 1;  // basis vectors final before this step; the step finalises column j of H
            FEDD_TRY(apply_B(u, wt));
            {
                ScopedTimer t(c, FEDD_T_ORTHO);
                const int ncg = (k + 1 + MD2_CG - 1) / MD2_CG;
                ScopedTimer td(c, FEDD_T_GS_DOT);
                td.bytes(8.0 * (double)n * (k + 2));   // k basis columns, u, B u
                if (md2_nch == 2)
                    hipLaunchKernelGGL(k_multidot2<2>, dim3(nblkd, std::min(md2_gy, ncg)), blk, 0, st, (const double*)V, ldv, n,
                                       k, (const double*)u, (const double*)wt, c->d_part.p, nblkd);
                else
                    hipLaunchKernelGGL(k_multidot2<4>, dim3(nblkd, std::min(md2_gy, ncg)), blk, 0, st, (const double*)V, ldv, n,
                                       k, (const double*)u, (const double*)wt, c->d_part.p, nblkd);
                td.stop();
                hipLaunchKernelGGL(k_reduce_cols, dim3(2 * k + 2), blk, 0, st, (const double*)c->d_part.p, S + o2.st, nblkd,
                                   (const int32_t*)nullptr);
                FEDD_TRY(allreduce_sum(c, S + o2.st, 2 * k + 2));
                hipLaunchKernelGGL(k_dcgs2_small, dim3(1), blk, (size_t)(6 * m + 8) * sizeof(double), st, S, o, o2, k, m,
                                   c->h_pinned_dev ? c->h_pinned_dev + 4 * (j & 1) : (double*)nullptr);
                if (k < m) {  // the last step of a cycle needs no further basis vector
                    ScopedTimer tu(c, FEDD_T_GS_UPDATE);
                    tu.bytes(8.0 * (double)n * (k + 4));   // k basis columns, u and B u in, v_{k+1} and the next u out
                    hipLaunchKernelGGL(k_axpy2, dim3(nblk2), blk, 0, st, V, ldv, n, k, (const double*)(S + o2.cf), m, u,
                                       (const double*)wt);
                    tu.stop();
                }
                t.stop();
            }
            if (!c->h_pinned_dev)
                FEDD_HIP(hipMemcpyAsync(c->h_pinned + 4 * (j & 1), S + o.misc, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
            FEDD_HIP(hipEventRecord(ev[j & 1], st));
            ++issued;
            ++queued;
            if (j > 0) {
                const int rc = check(j - 1);
                FEDD_CHECK(rc >= 0, "gmres: waiting for iteration %d failed", j - 1);
                if (rc) {
                    converged = rc == 1;
                    broke = rc == 2;
                    break;
                }
            }
        }
        if (!converged && !broke && checked < queued) {
            const int rc = check(queued - 1);
            FEDD_CHECK(rc >= 0, "gmres: waiting for iteration %d failed", queued - 1);
            converged = rc == 1;
            broke = rc == 2;
        }
        // x += M^-1 (V y) with the kfin finalised columns
        hipLaunchKernelGGL(k_backsolve, dim3(1), dim3(256), (size_t)(kfin + 1) * sizeof(double), st, S, o, kfin, m);
        hipLaunchKernelGGL(k_combine, dim3((unsigned)((n + AX_ROWS - 1) / AX_ROWS)), blk, 0, st, (const double*)V, ldv, n, kfin, (const double*)(S + o.y), r);
        if (use_prec) {
            FEDD_TRY(schwarz_apply(c, r, z, true));
            if (mk) hipLaunchKernelGGL(k_mask_mix, gn, blk, 0, st, mk, (const double*)z, (const double*)r, z, n);
            hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, (const double*)d_x, 1.0, (const double*)z, d_x, n);
        } else {
            hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, (const double*)d_x, 1.0, (const double*)r, d_x, n);
        }
        if (!converged && (its < max_it || broke)) {
            FEDD_TRY(spmv_owned(c, d_x, r));
            if (mk) hipLaunchKernelGGL(k_mask_mix, gn, blk, 0, st, mk, (const double*)r, (const double*)d_x, r, n);
            hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, d_b, -1.0, (const double*)r, r, n);
            FEDD_TRY(norm2_into(r, S + o.nrm + 3));
            if (broke) {   // rare path: the host reads the true residual (every rank takes the same decision)
                FEDD_HIP(hipMemcpyAsync(c->h_pinned, S + o.nrm + 3, sizeof(double), hipMemcpyDeviceToHost, st));
                FEDD_HIP(hipStreamSynchronize(st));
                relres = std::sqrt(std::max(c->h_pinned[0], 0.0)) / beta0;
                if (relres <= rtol) converged = true;
            }
        }
    }
    FEDD_HIP(hipGetLastError());
    FEDD_HIP(hipStreamSynchronize(st));
    if (its_out) *its_out = its;
    if (relres_out) *relres_out = relres;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// s-step GMRES (gmres_kind 2).  The Krylov basis is extended s vectors at a time: w_i = B w_{i-1},
// w_0 = q_{k-1} (monomial block basis), and the block W = [w_1 .. w_s] is orthogonalised against the k
// final basis vectors and within itself by block classical Gram-Schmidt with the Pythagorean inner
// product, twice (BCGS-PIP2; Carson, Lund, Rozloznik, Thomas, "Block Gram-Schmidt algorithms and their
// stability properties", 2022):
//   pass:  [C; G] = [Q_k W]^T W  (ONE sweep over the basis, one reduction);  R^T R = G - C^T C (Cholesky of an
//          s x s matrix);  W <- (W - Q_k C) R^-1  (one sweep).
// Two passes = four sweeps over the basis per s iterations instead of two per iteration, and two all-reduces per
// block instead of one per iteration.  With W = Q_k C + Q_new R (C = C1 + C2 R1, R = R2 R1) the Hessenberg
// columns follow from B q_{k-1} = w_1 and B w_i = w_{i+1}:
//   column k-1        = [C(:,0); R(0,0)]
//   columns k..k+s-2  = ([C(:,1:s); R(:,1:s)] - [Hbar_k C(:,0:s-1); 0]) R(0:s-1,0:s-1)^-1
// (Hoemmen, "Communication-avoiding Krylov subspace methods", 2010, section 3.3).  In exact arithmetic the
// iterates are those of standard GMRES.  A block whose Cholesky factorisation meets a pivot below the
// threshold is cut there (the columns before it are valid) and the following blocks are shorter; the
// convergence claim of the recurrence is checked against the true residual before it is accepted.

// basis columns per group of the block dot kernel: groups x block columns = the wave totals one transpose reduction yields
constexpr int ss_cg(int S) { return S >= 16 ? 2 : 4; }

// exchange step of the transpose reduction between lanes l and l ^ OFF for one pair of values: afterwards `lo` holds, in
// the lanes with bit OFF clear, lo(l) + lo(l ^ OFF) and, in the lanes with it set, hi(l) + hi(l ^ OFF).
// OFF = 32 / 16: gfx950's v_permlane32_swap / v_permlane16_swap move the halves (rows 2-3 of the first operand <-> rows
// 0-1 of the second; odd rows of the first <-> even rows of the second) without going through the LDS crossbar.
template <int OFF>
__device__ __forceinline__ double xchg_add(double lo, double hi, int lane) {
    if constexpr (OFF == 32 || OFF == 16) {
        const unsigned l0 = __double2loint(lo), l1 = __double2hiint(lo), h0 = __double2loint(hi), h1 = __double2hiint(hi);
        if constexpr (OFF == 32) {
            const auto r0 = __builtin_amdgcn_permlane32_swap(l0, h0, false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(l1, h1, false, false);
            return __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
        } else {
            const auto r0 = __builtin_amdgcn_permlane16_swap(l0, h0, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(l1, h1, false, false);
            return __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
        }
    } else {
        const bool up = (lane & OFF) != 0;
        const double keep = up ? hi : lo;
        const double send = up ? lo : hi;
        return keep + __shfl_xor(send, OFF, 64);
    }
}

// all-lanes transpose reduction of NV values: afterwards lane l holds the wave total of value l >> (6 - log2 NV)
// (values first, then selects on values: a select between two array ELEMENTS becomes an indexed access and sends the whole
// array to scratch memory)
template <int NV, int OFF, int W>
__device__ __forceinline__ void wave_reduce_step(double (&a)[NV], int lane) {
    if constexpr (W > 1) {
#pragma unroll
        for (int i = 0; i < W / 2; ++i) a[i] = xchg_add<OFF>(a[i], a[i + W / 2], lane);
    } else {
        a[0] += __shfl_xor(a[0], OFF, 64);
    }
    if constexpr (OFF > 1) wave_reduce_step<NV, OFF / 2, (W > 1 ? W / 2 : 1)>(a, lane);
}
template <int NV>
__device__ __forceinline__ double wave_reduce_transpose(double (&a)[NV], int lane) {
    wave_reduce_step<NV, 32, NV>(a, lane);
    return a[0];
}

// branch-free 16-byte load of rows (r, r + 1) of a basis column (columns are padded to ldv >= n + (n & 1), so the pair
// at a clamped row is always readable; rows past n read as zero through selects, never through arithmetic on padding)
struct RowPair {
    int64_t rc;   // clamped row
    bool v0, v1;
};
__device__ __forceinline__ RowPair row_pair(int64_t r, int64_t n) {
    RowPair q;
    q.v0 = r < n;
    q.v1 = r + 1 < n;
    q.rc = q.v0 ? r : 0;
    return q;
}
template <bool NT>
__device__ __forceinline__ double2 ldraw(const double* __restrict__ p, const RowPair& q) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d t = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + q.rc)) : *reinterpret_cast<const v2d*>(p + q.rc);
    return double2{t.x, t.y};
}
template <bool NT>
__device__ __forceinline__ double2 ldp(const double* __restrict__ p, const RowPair& q) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d t = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + q.rc)) : *reinterpret_cast<const v2d*>(p + q.rc);
    double2 v;
    v.x = q.v0 ? t.x : 0.0;
    v.y = q.v1 ? t.y : 0.0;
    return v;
}

constexpr int SS_LDS_GROUPS = 32;   // column groups whose wave totals are parked in LDS between two workgroup barriers

// partial[(col * S + j) * nblk + blk] = sum over the workgroup's rows of V_col . W_j, col < k + sa
// (W_j = V_{k+j}); the W tile stays in registers across the column groups, every basis column is read once.
// The wave totals of a column group are parked in LDS and added (fixed order) once per SS_LDS_GROUPS groups:
// no barrier between the loads of consecutive groups.
// HV = 2 ("split"): the two halves of the workgroup take the SAME rows and one half of the block's columns each (S / 2 per
// lane): a basis value is then loaded by two lanes of the workgroup (the second finds it in the L1), but a lane's register
// tile and reduction are those of the 8-column kernel, which runs at 0.69 of peak where the 16-column tile (232 VGPRs, two
// FMAs and two exchanges per byte) reaches 0.59.
template <int S, int NCH, int SS_CG, int HV = 1>
__global__ __launch_bounds__(256) void k_blockdot(const double* __restrict__ V, int64_t ldv, int64_t n, int k, int sa_req,
                                                  const int32_t* __restrict__ d_sa, double* __restrict__ partial, int nblk) {
    constexpr int SW = S / HV;              // block columns per lane
    constexpr int NV = SS_CG * SW;
    constexpr int LH = 256 / HV;            // lanes of a half
    constexpr int WPH = 4 / HV;             // waves of a half
    static_assert(NV == 16 || NV == 32 || NV == 64, "block size");
    __shared__ double sh[SS_LDS_GROUPS][4][NV];
    const int sa = d_sa ? min(*d_sa, sa_req) : sa_req;
    if (sa <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = tid / LH, th = tid % LH;
    RowPair rp[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
        rp[ch] = row_pair((int64_t)blockIdx.x * (2 * LH * NCH) + 2 * LH * ch + 2 * th, n);
    double2 w[SW][NCH];
#pragma unroll
    for (int j = 0; j < SW; ++j)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            // (columns past sa: any readable column, zeroed by the select)
            const int jj = half * SW + j;
            const double2 t = ldp<false>(V + (int64_t)(k + (jj < sa ? jj : 0)) * ldv, rp[ch]);
            w[j][ch].x = jj < sa ? t.x : 0.0;
            w[j][ch].y = jj < sa ? t.y : 0.0;
        }
    const int ncol = k + sa;
    const int ncg = (ncol + SS_CG - 1) / SS_CG;
    int parked = 0, first_cg = blockIdx.y;
    constexpr int GRP = 64 / NV;
    auto flush = [&](int count) {
        __syncthreads();
        for (int e = tid; e < count * NV * HV; e += 256) {
            const int g = e / (NV * HV), rem = e % (NV * HV), hv = rem / NV, idx = rem % NV;
            const int col = (first_cg + g * (int)gridDim.y) * SS_CG + idx / SW;
            if (col < ncol) {
                double tsum = sh[g][hv * WPH][idx];
#pragma unroll
                for (int q = 1; q < WPH; ++q) tsum += sh[g][hv * WPH + q][idx];
                partial[((int64_t)col * S + hv * SW + (idx % SW)) * nblk + blockIdx.x] = tsum;
            }
        }
        __syncthreads();
    };
    double2 v[SS_CG][NCH];
    auto load_group = [&](int cg) {
#pragma unroll
        for (int cc = 0; cc < SS_CG; ++cc) {
            const int col = cg * SS_CG + cc;
            const double* __restrict__ a = V + (int64_t)(col < ncol ? col : 0) * ldv;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                // final basis columns are streamed (non-temporal); the block's own columns are re-read soon.
                // Raw loads, no selects behind them (a select would make the wave wait for the load right here instead of
                // at the products after the reduction of the previous group): rows past n meet w = 0 there (the padding
                // rows of the basis are zeroed when it is allocated), columns past ncol are dropped when the sums are stored
                v[cc][ch] = col < k ? ldraw<true>(a, rp[ch]) : ldraw<false>(a, rp[ch]);
            }
        }
    };
    if ((int)blockIdx.y < ncg) load_group(blockIdx.y);
    for (int cg = blockIdx.y; cg < ncg; cg += gridDim.y) {
        double acc[NV];
#pragma unroll
        for (int cc = 0; cc < SS_CG; ++cc)
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                double s = 0.0;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) s += v[cc][ch].x * w[j][ch].x + v[cc][ch].y * w[j][ch].y;
                acc[cc * SW + j] = s;
            }
        // the next group's loads are in flight while this one is reduced
        if (cg + (int)gridDim.y < ncg) load_group(cg + gridDim.y);
        const double tot = wave_reduce_transpose<NV>(acc, lane);
        if ((lane & (GRP - 1)) == 0) sh[parked][wave][lane / GRP] = tot;
        if (++parked == SS_LDS_GROUPS) {
            flush(parked);
            parked = 0;
            first_cg = cg + gridDim.y;
        }
    }
    if (parked) flush(parked);
}

// W <- (W - V_k C) Rinv : cf = C [k][S] row-wise, rinv [S][S] row-wise upper triangular; columns >= sa untouched
// COMB (the block that fills a restart cycle, second update): the sweep also forms  comb = sum_{c < k + sa - 1} y_c V_c  with
// the finished columns of the block -- the combination k_combine would read the whole basis for right afterwards (y from
// k_backsolve over all k - 1 + sa Hessenberg columns; only valid if the block is not cut, which the host knows later)
template <int S, bool COMB = false>
__global__ __launch_bounds__(256) void k_blockaxpy(double* __restrict__ V, int64_t ldv, int64_t n, int k, int sa_req,
                                                   const int32_t* __restrict__ d_sa, const double* __restrict__ cf,
                                                   const double* __restrict__ rinv, const double* __restrict__ yv = nullptr,
                                                   double* __restrict__ comb = nullptr) {
    const int sa = min(*d_sa, sa_req);
    if (sa <= 0) return;
    const int tid = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * AX_ROWS + 2 * tid;
    const RowPair rp = row_pair(r, n);
    double2 cb = double2{0.0, 0.0};
    double2 acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = double2{0.0, 0.0};
    int c = 0;
    for (; c + 8 <= k; c += 8) {
        double2 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = ldp<true>(V + (int64_t)(c + u) * ldv, rp);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const double cc = cf[(c + u) * S + j];   // wave-uniform: scalar loads
                acc[j].x += cc * q[u].x;
                acc[j].y += cc * q[u].y;
            }
            if constexpr (COMB) {
                const double yc = yv[c + u];
                cb.x += yc * q[u].x;
                cb.y += yc * q[u].y;
            }
        }
    }
    for (; c < k; ++c) {
        const double2 q = ldp<true>(V + (int64_t)c * ldv, rp);
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const double cc = cf[c * S + j];
            acc[j].x += cc * q.x;
            acc[j].y += cc * q.y;
        }
        if constexpr (COMB) {
            const double yc = yv[c];
            cb.x += yc * q.x;
            cb.y += yc * q.y;
        }
    }
    double2 t[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const double2 wv = ldp<false>(V + (int64_t)(k + (j < sa ? j : 0)) * ldv, rp);
        t[j].x = j < sa ? wv.x - acc[j].x : 0.0;
        t[j].y = j < sa ? wv.y - acc[j].y : 0.0;
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
        if (j < sa) {
            double2 o = double2{0.0, 0.0};
#pragma unroll
            for (int i = 0; i <= j; ++i) {
                const double ri = rinv[i * S + j];
                o.x += ri * t[i].x;
                o.y += ri * t[i].y;
            }
            double* wj = V + (int64_t)(k + j) * ldv;
            if (rp.v1) *reinterpret_cast<double2*>(wj + r) = o;
            else if (rp.v0) wj[r] = o.x;
            if constexpr (COMB) {
                if (j < sa - 1) {       // (the last basis vector takes no part in the solution update)
                    const double yc = yv[k + j];
                    cb.x += yc * o.x;
                    cb.y += yc * o.y;
                }
            }
        }
    }
    if constexpr (COMB) {
        if (rp.v1) *reinterpret_cast<double2*>(comb + r) = cb;
        else if (rp.v0) comb[r] = cb.x;
    }
}

// First update and second dot of the block Gram-Schmidt in ONE sweep (option "gmres_fuse"): a lane takes a row pair, forms
//   W' = (W - V_k C1) R1^-1   (k_blockaxpy: all k basis columns read once, W' stored)
// and then, with W' still in registers, the second pass' products  [V_k W']^T W'  (k_blockdot) over the same rows: the
// basis columns are read a second time -- from the Infinity Cache, where the first read of the workgroup's rows left them
// (tools/microbench/reread.hip: two passes over the same rows of 64 columns 1.43 ms against 2 x 0.84 for two sweeps).
// partial[(col * S + j) * nblk + blk] as k_blockdot writes it, nblk = gridDim.x (512 rows per workgroup).
template <int S>
__global__ __launch_bounds__(256) void k_blockfuse(double* __restrict__ V, int64_t ldv, int64_t n, int k, int sa_req,
                                                   const int32_t* __restrict__ d_sa, const double* __restrict__ cf,
                                                   const double* __restrict__ rinv, double* __restrict__ partial, int nblk) {
    constexpr int SS_CG = 2, NV = SS_CG * S;
    static_assert(NV == 32, "block size");
    __shared__ double sh[SS_LDS_GROUPS][4][NV];
    const int sa = min(*d_sa, sa_req);
    if (sa <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r = (int64_t)blockIdx.x * AX_ROWS + 2 * tid;
    const RowPair rp = row_pair(r, n);
    double2 w[S];
    {
        double2 acc[S];
#pragma unroll
        for (int j = 0; j < S; ++j) acc[j] = double2{0.0, 0.0};
        int c = 0;
        for (; c + 8 <= k; c += 8) {
            double2 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = ldp<false>(V + (int64_t)(c + u) * ldv, rp);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    const double cc = cf[(c + u) * S + j];   // wave-uniform: scalar loads
                    acc[j].x += cc * q[u].x;
                    acc[j].y += cc * q[u].y;
                }
        }
        for (; c < k; ++c) {
            const double2 q = ldp<false>(V + (int64_t)c * ldv, rp);
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const double cc = cf[c * S + j];
                acc[j].x += cc * q.x;
                acc[j].y += cc * q.y;
            }
        }
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const double2 wv = ldp<false>(V + (int64_t)(k + (j < sa ? j : 0)) * ldv, rp);
            acc[j].x = j < sa ? wv.x - acc[j].x : 0.0;
            acc[j].y = j < sa ? wv.y - acc[j].y : 0.0;
        }
#pragma unroll
        for (int j = 0; j < S; ++j) {
            double2 o = double2{0.0, 0.0};
            if (j < sa) {
#pragma unroll
                for (int i = 0; i <= j; ++i) {
                    const double ri = rinv[i * S + j];
                    o.x += ri * acc[i].x;
                    o.y += ri * acc[i].y;
                }
                double* wj = V + (int64_t)(k + j) * ldv;
                if (rp.v1) *reinterpret_cast<double2*>(wj + r) = o;
                else if (rp.v0) wj[r] = o.x;
            }
            // (rows past n: every value above came through ldp's zeros, so o is 0 there)
            w[j] = o;
        }
    }
    // ---- second pass' dot products over the same rows ----
    const int ncol = k + sa;
    const int ncg = (ncol + SS_CG - 1) / SS_CG;
    int parked = 0, first_cg = 0;
    constexpr int GRP = 64 / NV;
    auto flush = [&](int count) {
        __syncthreads();
        for (int e = tid; e < count * NV; e += 256) {
            const int g = e / NV, idx = e % NV;
            const int col = (first_cg + g) * SS_CG + idx / S;
            if (col < ncol) partial[((int64_t)col * S + (idx % S)) * nblk + blockIdx.x] = ((sh[g][0][idx] + sh[g][1][idx]) + sh[g][2][idx]) + sh[g][3][idx];
        }
        __syncthreads();
    };
    double2 v[SS_CG];
    auto load_group = [&](int cg) {
#pragma unroll
        for (int cc = 0; cc < SS_CG; ++cc) {
            const int col = cg * SS_CG + cc;
            const double* __restrict__ a = V + (int64_t)(col < ncol ? col : 0) * ldv;
            // basis columns: second and last read of this sweep (non-temporal); the block's own columns were stored above by this lane
            v[cc] = col < k ? ldraw<true>(a, rp) : ldp<false>(a, rp);
        }
    };
    load_group(0);
    for (int cg = 0; cg < ncg; ++cg) {
        double acc[NV];
#pragma unroll
        for (int cc = 0; cc < SS_CG; ++cc)
#pragma unroll
            for (int j = 0; j < S; ++j) acc[cc * S + j] = v[cc].x * w[j].x + v[cc].y * w[j].y;
        if (cg + 1 < ncg) load_group(cg + 1);
        const double tot = wave_reduce_transpose<NV>(acc, lane);
        if ((lane & (GRP - 1)) == 0) sh[parked][wave][lane / GRP] = tot;
        if (++parked == SS_LDS_GROUPS) {
            flush(parked);
            parked = 0;
            first_cg = cg + 1;
        }
    }
    if (parked) flush(parked);
}

struct Off3 {
    int Hraw, P, C1, R1, R1i, Cc, cf1, ri1, cf2, ri2, th, res;
};

// Upper Cholesky factor of the sa x sa leading block of the symmetric A (upper part read) via the unit-diagonal scaling, cut
// at the first column whose pivot (relative to gdiag: the squared sine of the angle between the column and the span of
// everything before it) is <= tol.  Returns the accepted columns; R and its inverse Ri are zero outside the accepted upper
// triangle.  Called by the whole workgroup (>= S * S lanes), A / W / R / Ri / dd in LDS; right-looking, one barrier-separated
// step per column.
template <int S>
__device__ int chol_cut(double (&A)[S][S], const double* gdiag, int sa, double tol, double (&W)[S][S], double (&R)[S][S],
                        double (&Ri)[S][S], double* dd, int* s_ok, int tid) {
    const int i = tid / S, j = tid % S;
    if (tid < S) dd[tid] = (tid < sa && A[tid][tid] > 0.0) ? sqrt(A[tid][tid]) : 0.0;
    if (tid == 0) *s_ok = sa;
    __syncthreads();
    if (tid < S * S) {
        W[i][j] = (i <= j && dd[i] > 0.0 && dd[j] > 0.0) ? A[i][j] / (dd[i] * dd[j]) : 0.0;
        R[i][j] = 0.0;
        Ri[i][j] = 0.0;
    }
    __syncthreads();
    for (int c = 0; c < sa; ++c) {
        if (tid == 0) {
            const double piv = W[c][c];
            if (!(dd[c] > 0.0) || !(piv > 0.0) || !(piv * A[c][c] > tol * gdiag[c])) *s_ok = c;
        }
        __syncthreads();
        if (*s_ok <= c) break;   // (uniform: read after the barrier)
        const double rp = 1.0 / sqrt(W[c][c]);
        __syncthreads();
        if (tid < S && tid >= c) R[c][tid] = W[c][tid] * rp;   // row c of U (R[c][c] = sqrt(piv))
        __syncthreads();
        if (tid < S * S && i > c && i <= j) W[i][j] -= R[c][i] * R[c][j];
        __syncthreads();
    }
    const int ok = *s_ok;
    if (tid < S * S) R[i][j] = (i <= j && j < ok) ? R[i][j] * dd[j] : 0.0;
    __syncthreads();
    if (tid < ok) {   // column tid of the inverse by back substitution
        const int cj = tid;
        Ri[cj][cj] = 1.0 / R[cj][cj];
        for (int r = cj - 1; r >= 0; --r) {
            double acc = 0.0;
            for (int p = r + 1; p <= cj; ++p) acc += R[r][p] * Ri[p][cj];
            Ri[r][cj] = -acc / R[r][r];
        }
    }
    __syncthreads();
    return ok;
}

// S1 = G - C^T C from the reduced dot products P [(k + sa)][S]
template <int S>
__device__ void gram_minus(const double* __restrict__ P, int k, double (&G)[S][S], double (&A)[S][S], int tid) {
    if (tid < S * S) {
        const int i = tid / S, j = tid % S;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int c = 0;
        for (; c + 4 <= k; c += 4) {
            s0 += P[(c + 0) * S + i] * P[(c + 0) * S + j];
            s1 += P[(c + 1) * S + i] * P[(c + 1) * S + j];
            s2 += P[(c + 2) * S + i] * P[(c + 2) * S + j];
            s3 += P[(c + 3) * S + i] * P[(c + 3) * S + j];
        }
        for (; c < k; ++c) s0 += P[c * S + i] * P[c * S + j];
        const double g = P[(k + i) * S + j];
        G[i][j] = g;
        A[i][j] = g - ((s0 + s1) + (s2 + s3));
    }
}

// pass 1 of a block: Cholesky of G - C1^T C1 (cut where the block basis becomes dependent), coefficients of sweep 2
template <int S>
__global__ __launch_bounds__(256) void k_ss_pass1(double* __restrict__ Sx, Off3 o3, int k, int sa_req, double tol,
                                                  int32_t* __restrict__ d_sa) {
    __shared__ double G[S][S], A[S][S], W[S][S], R[S][S], Ri[S][S];
    __shared__ double gd[S], dd[S];
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    const double* P = Sx + o3.P;
    gram_minus<S>(P, k, G, A, tid);
    __syncthreads();
    if (tid < S) gd[tid] = G[tid][tid];
    __syncthreads();
    const int ok1 = chol_cut<S>(A, gd, sa_req, tol, W, R, Ri, dd, &s_ok, tid);
    if (tid == 0) d_sa[0] = ok1;
    __syncthreads();
    for (int i = tid; i < k * S; i += 256) {
        const double v = P[i];
        Sx[o3.C1 + i] = v;
        Sx[o3.cf1 + i] = v;
    }
    if (tid < S * S) {
        Sx[o3.R1 + tid] = R[tid / S][tid % S];
        Sx[o3.R1i + tid] = Ri[tid / S][tid % S];
        Sx[o3.ri1 + tid] = Ri[tid / S][tid % S];
    }
}

// pass 2 of a block: second Cholesky, combined coefficients, the new Hessenberg columns (formulas above), their
// Givens rotations and the residual after each of them.
// host_out: [0] = accepted columns (0: breakdown, the column k-1 is still final), [1] = 1 if breakdown, [2 + i] =
// residual after column k-1+i
template <int S>
__global__ __launch_bounds__(256) void k_ss_pass2(double* __restrict__ Sx, Off o, Off3 o3, int k, int sa_req, int m, double tol,
                                                  int32_t* __restrict__ d_sa, double* __restrict__ host_out) {
    extern __shared__ double dyn[];   // hc[S][m+2] | lcs[m] | lsn[m]
    __shared__ double G[S][S], A[S][S], W[S][S], R2[S][S], R2i[S][S], R1[S][S], R1i[S][S], Rc[S][S], Rci[S][S];
    __shared__ double gd[S], th[S], dd[S];
    __shared__ int s_sa;
    double* hc = dyn;
    double* lcs = hc + (size_t)S * (m + 2);
    double* lsn = lcs + m;
    const int tid = threadIdx.x, ldh = m + 1;
    const double* P = Sx + o3.P;
    double* Hraw = Sx + o3.Hraw;
    const int sa1 = min(d_sa[0], sa_req);
    if (tid < S * S) {
        R1[tid / S][tid % S] = Sx[o3.R1 + tid];
        R1i[tid / S][tid % S] = Sx[o3.R1i + tid];
    }
    if (tid < S) th[tid] = Sx[o3.th + tid];
    for (int i = tid; i < k - 1; i += 256) {
        lcs[i] = Sx[o.cs + i];
        lsn[i] = Sx[o.sn + i];
    }
    int sa = 0;
    if (sa1 > 0) {
        gram_minus<S>(P, k, G, A, tid);
        __syncthreads();
        if (tid < S) gd[tid] = G[tid][tid];
        __syncthreads();
        sa = chol_cut<S>(A, gd, sa1, tol, W, R2, R2i, dd, &s_sa, tid);
    }
    __syncthreads();
    const bool brk = sa == 0;
    if (!brk) {
        if (tid < S * S) {
            const int i = tid / S, j = tid % S;
            double s = 0.0, t = 0.0;
            for (int p = 0; p < S; ++p) {
                s += R2[i][p] * R1[p][j];
                t += R1i[i][p] * R2i[p][j];
            }
            Rc[i][j] = (i < sa && j < sa) ? s : 0.0;
            Rci[i][j] = (i < sa && j < sa) ? t : 0.0;
        }
        // combined C = C1 + C2 R1 ; sweep-4 coefficients C2
        for (int idx = tid; idx < k * S; idx += 256) {
            const int cc = idx / S, j = idx % S;
            double s = Sx[o3.C1 + idx];
            for (int i = 0; i <= j; ++i) s += P[cc * S + i] * R1[i][j];
            Sx[o3.Cc + idx] = s;
            Sx[o3.cf2 + idx] = P[idx];
        }
        if (tid < S * S) Sx[o3.ri2 + tid] = R2i[tid / S][tid % S];
    } else {
        // breakdown: w_1 lies in the span of the basis (to the threshold): column k-1 = [C1(:,0); 0]
        for (int idx = tid; idx < k * S; idx += 256) Sx[o3.Cc + idx] = Sx[o3.C1 + idx];
        if (tid < S * S) Rc[tid / S][tid % S] = 0.0;
    }
    __syncthreads();
    // column k-1
    for (int cidx = tid; cidx <= k; cidx += 256) {
        const double v = cidx < k ? Sx[o3.Cc + cidx * S] + (cidx == k - 1 ? th[0] : 0.0) : Rc[0][0];
        Hraw[(int64_t)(k - 1) * ldh + cidx] = v;
    }
    __syncthreads();
    // columns k .. k+sa-2
    const int nc = brk ? 0 : sa - 1;
    if (nc > 0) {
        for (int r = tid; r < k + sa; r += 256) {
            double x[S];
#pragma unroll
            for (int cidx = 0; cidx < S; ++cidx) x[cidx] = 0.0;
            // y = (Hbar_k C)(r, :): the same q for every lane, so that the coefficients C(q, :) are wave-uniform (scalar loads)
            // and only Hbar(r, q) is a per-lane load (coalesced over r); entries below the sub-diagonal are zero by structure
            // (with a per-lane loop start every lane fetched its own 15 coefficients per step: 89 us per block at k ~ 60)
            // (eight entries of the Hessenberg row requested before any is used: one global round trip per eight columns instead
            // of one per column -- the kernel is a single workgroup, nothing else hides that latency)
            for (int q0 = 0; q0 < k; q0 += 8) {
                double h[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = q0 + u;
                    h[u] = (q < k && r <= k && q + 1 >= r) ? Hraw[(int64_t)q * ldh + r] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = q0 + u;
                    if (q < k) {
#pragma unroll
                        for (int cidx = 0; cidx < S - 1; ++cidx) x[cidx] -= h[u] * Sx[o3.Cc + q * S + cidx];
                    }
                }
            }
#pragma unroll
            for (int cidx = 0; cidx < S - 1; ++cidx) {
                if (cidx < nc) {
                    if (r < k) x[cidx] += Sx[o3.Cc + r * S + cidx + 1] + th[cidx + 1] * Sx[o3.Cc + r * S + cidx];
                    else x[cidx] += Rc[r - k][cidx + 1] + th[cidx + 1] * Rc[r - k][cidx];
                }
            }
#pragma unroll
            for (int cidx = 0; cidx < S - 1; ++cidx) {
                if (cidx < nc && r <= k + cidx + 1) {
                    double hv = 0.0;
#pragma unroll
                    for (int cp = 0; cp < S - 1; ++cp)
                        if (cp <= cidx) hv += x[cp] * Rci[cp][cidx];
                    Hraw[(int64_t)(k + cidx) * ldh + r] = hv;
                }
            }
        }
    }
    __syncthreads();
    // Givens.  The rotations of the earlier columns (q < k - 1) are applied to all new columns at once, a lane per column (each
    // chain is sequential in q, the columns are independent); then lane 0 runs the triangular rest -- the rotations the block
    // itself creates -- column by column.  Per column the same operations in the same order as one chain after the other
    // (which took 16 x k dependent LDS steps on one lane: 67 of the kernel's 95 us at k ~ 60).
    const int ncolH = brk ? 1 : sa;
    for (int i = 0; i < ncolH; ++i) {
        const int jj = k - 1 + i;
        double* hci = hc + (size_t)i * (m + 2);
        for (int q = tid; q <= jj + 1; q += 256) hci[q] = Hraw[(int64_t)jj * ldh + q];
    }
    __syncthreads();
    if (tid < ncolH) {
        double* hci = hc + (size_t)tid * (m + 2);
        for (int q = 0; q < k - 1; ++q) {
            const double a = lcs[q] * hci[q] + lsn[q] * hci[q + 1];
            hci[q + 1] = -lsn[q] * hci[q] + lcs[q] * hci[q + 1];
            hci[q] = a;
        }
    }
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < ncolH; ++i) {
            const int jj = k - 1 + i;
            double* hci = hc + (size_t)i * (m + 2);
            for (int q = k - 1; q < jj; ++q) {
                const double a = lcs[q] * hci[q] + lsn[q] * hci[q + 1];
                hci[q + 1] = -lsn[q] * hci[q] + lcs[q] * hci[q + 1];
                hci[q] = a;
            }
            const double d = hypot(hci[jj], hci[jj + 1]);
            const double cj = d > 0 ? hci[jj] / d : 1.0, sj = d > 0 ? hci[jj + 1] / d : 0.0;
            lcs[jj] = cj;
            lsn[jj] = sj;
            Sx[o.cs + jj] = cj;
            Sx[o.sn + jj] = sj;
            hci[jj] = d;
            hci[jj + 1] = 0.0;
            const double gj = Sx[o.g + jj];
            Sx[o.g + jj + 1] = -sj * gj;
            Sx[o.g + jj] = cj * gj;
            Sx[o3.res + i] = fabs(sj * gj);
        }
    }
    __syncthreads();
    for (int i = 0; i < ncolH; ++i) {
        const int jj = k - 1 + i;
        const double* hci = hc + (size_t)i * (m + 2);
        double* H = Sx + o.H + (int64_t)jj * ldh;
        for (int q = tid; q <= jj + 1; q += 256) H[q] = hci[q];
    }
    __syncthreads();
    if (tid == 0) {
        d_sa[0] = sa;
        Sx[o.misc + 4] = (double)sa;
        Sx[o.misc + 5] = brk ? 1.0 : 0.0;
        if (host_out) {
            host_out[0] = (double)sa;
            host_out[1] = brk ? 1.0 : 0.0;
            for (int i = 0; i < S; ++i) host_out[2 + i] = i < ncolH ? Sx[o3.res + i] : 0.0;
            __threadfence_system();
        }
    }
}

__global__ void k_ss_cycle_init(double* __restrict__ S, Off o, Off3 o3, int m, int s, const double* __restrict__ rr) {
    const double beta = sqrt(rr[0]);
    for (int i = threadIdx.x; i <= m; i += blockDim.x) S[o.g + i] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        S[o.g] = beta;
        S[o.misc + 1] = beta > 0 ? 1.0 / beta : 0.0;
        S[o.misc + 3] = beta;
    }
}

// Real parts of the eigenvalues of a small upper Hessenberg matrix (column-major, leading dimension ld) by the unshifted QR
// iteration with Givens rotations: crude (linear convergence), but the Ritz values only have to place the shifts of the Newton
// block basis over the spectrum; whatever 2 x 2 blocks are left on the diagonal are resolved in closed form.
static void hessenberg_real_parts(std::vector<double> H, int n, int ld, double* out) {
    std::vector<double> cs((size_t)n), sn((size_t)n);
    auto at = [&](int i, int j) -> double& { return H[(size_t)j * ld + i]; };
    for (int sweep = 0; sweep < 500; ++sweep) {
        for (int i = 0; i + 1 < n; ++i) {   // H <- Q^T H
            const double a = at(i, i), b = at(i + 1, i), d = std::hypot(a, b);
            cs[i] = d > 0 ? a / d : 1.0;
            sn[i] = d > 0 ? b / d : 0.0;
            for (int j = i; j < n; ++j) {
                const double u = at(i, j), v = at(i + 1, j);
                at(i, j) = cs[i] * u + sn[i] * v;
                at(i + 1, j) = -sn[i] * u + cs[i] * v;
            }
        }
        for (int i = 0; i + 1 < n; ++i)     // H <- H Q
            for (int r = 0; r <= std::min(i + 1, n - 1); ++r) {
                const double u = at(r, i), v = at(r, i + 1);
                at(r, i) = cs[i] * u + sn[i] * v;
                at(r, i + 1) = -sn[i] * u + cs[i] * v;
            }
    }
    for (int i = 0; i < n;) {
        const double sub = i + 1 < n ? std::fabs(at(i + 1, i)) : 0.0;
        if (i + 1 < n && sub > 1e-10 * (std::fabs(at(i, i)) + std::fabs(at(i + 1, i + 1)))) {
            const double a = at(i, i), b = at(i, i + 1), c2 = at(i + 1, i), d = at(i + 1, i + 1);
            const double tr = 0.5 * (a + d), disc = 0.25 * (a - d) * (a - d) + b * c2;
            const double rt = disc > 0.0 ? std::sqrt(disc) : 0.0;   // complex pair: its real part, twice
            out[i] = tr + rt;
            out[i + 1] = tr - rt;
            i += 2;
        } else {
            out[i] = at(i, i);
            ++i;
        }
    }
}

// Leja ordering: the largest point first, then always the point that maximises the product of its distances to the points taken;
// every prefix of the sequence is then spread over the whole set (a short last block uses a prefix)
static void leja_order(double* v, int n) {
    std::vector<double> in(v, v + n), out;
    std::vector<char> used((size_t)n, 0);
    for (int t = 0; t < n; ++t) {
        int best = -1;
        double bestv = -1e300;
        for (int i = 0; i < n; ++i) {
            if (used[(size_t)i]) continue;
            double score = 0.0;
            if (out.empty()) score = std::fabs(in[(size_t)i]);
            else for (double o : out) score += std::log(std::max(std::fabs(in[(size_t)i] - o), 1e-300));
            if (score > bestv) {
                bestv = score;
                best = i;
            }
        }
        used[(size_t)best] = 1;
        out.push_back(in[(size_t)best]);
    }
    std::copy(out.begin(), out.end(), v);
}

template <int S>
static int gmres_solve_sstep(fedd_ctx* c, const double* d_b, double* d_x, double rtol, int max_it, int restart,
                             int use_prec, int* its_out, double* relres_out) {
    // c->gm_nr > 1: stacked vectors X[row * nr + j] (nr right-hand sides with one matrix: the GDSW extension solves); the
    // operator and the preconditioner are the stacked ones of multi.hip, everything else sees vectors nr times as long
    const int nr = c->gm_nr > 1 ? c->gm_nr : 1;
    FEDD_CHECK(nr == 1 || (nr == MULTI_NR && use_prec && multi_rhs_ok(c)), "gmres: stacked solve with %d right-hand sides", nr);
    const int64_t n = c->n_rows * nr;
    const int m = std::min(restart, max_it);
    const int64_t ldv = (n + 15) & ~(int64_t)15;
    FEDD_CHECK(m + 2 <= 1024, "gmres: restart length above 1022 is not supported");
    const int nblk = (int)((n + MD_ROWS - 1) / MD_ROWS), nblk2 = (int)((n + AX_ROWS - 1) / AX_ROWS);
    // block dot kernel: 512 NCH rows per workgroup, CG columns per transpose reduction (option "gmres_dotv": 0 = 2 chunks x ss_cg
    // columns; 1 = 1 chunk x ss_cg; 2 = 1 chunk x 2 ss_cg)
    const int dotv = c->gmres_dotv;
    // ("gmres_dotv" 3: S = 16 as two 8-column halves per workgroup over the same rows -- measured slower, 1.12 against 1.08 ms per
    // sweep at cfg 3: the second half's loads do not all hit the L1)
    const bool dot_split = S == 16 && dotv == 3;
    const int dot_nch = dot_split ? 2 : (dotv == 0 ? 2 : 1), dot_cg = dot_split ? 4 : (dotv == 2 ? std::min(64 / S, 2 * ss_cg(S)) : ss_cg(S));
    const int nblkd = dot_split ? (int)((n + 511) / 512) : (int)((n + 512 * dot_nch - 1) / (512 * dot_nch));
    {
        // the block kernels read whole 16-byte row pairs and rely on the padding rows [n, ldv) of every column being zero
        // (and on finite data everywhere): a freshly (re)allocated basis, or one last used with another vector length, is cleared
        const double* before = c->d_V.p;
        FEDD_TRY(c->d_V.ensure((size_t)(m + 1) * ldv));
        // (a buffer that another vector length used holds finite values everywhere: only its padding rows need the clearing)
        if (c->d_V.p != before) FEDD_HIP(hipMemsetAsync(c->d_V.p, 0, c->d_V.cap * sizeof(double), c->stream));
        else if (c->gm_V_ldv != ldv && ldv != n) FEDD_HIP(hipMemsetAsync(c->d_V.p, 0, (size_t)(m + 1) * ldv * sizeof(double), c->stream));
        c->gm_V_ldv = ldv;
    }
    const int64_t nc = (std::max<int64_t>(c->n_rows, c->n_cols) * nr + 15) & ~(int64_t)15;
    const bool ghosts = c->n_cols != c->n_rows || !c->halo.peers.empty();
    FEDD_TRY(c->d_w.ensure(std::max<size_t>((size_t)2 * nc, c->d_w.cap)));   // x trial | A x trial
    FEDD_TRY(c->d_Z.ensure((size_t)nc * (nr > 1 && ghosts ? 3 : 2)));      // (stacked, several ranks: + a copy with a ghost tail)
    FEDD_TRY(c->d_part.ensure(std::max((size_t)(m + 1 + S) * S * (c->gmres_fuse != 0 ? std::max(nblkd, nblk2) : nblkd), (size_t)std::max(nblk, nblk2))));
    FEDD_TRY(c->d_flags.ensure(16));
    Off o;
    Off3 o3;
    int p = 0;
    o.H = p; p += (m + 1) * m;
    o.cs = p; p += m;
    o.sn = p; p += m;
    o.g = p; p += m + 1;
    o.h1 = p; p += m + 2;
    o.h2 = p; p += m + 2;
    o.nrm = p; p += 4;
    o.y = p; p += m;
    o.misc = p; p += 8;
    o3.Hraw = p; p += (m + 1) * m;
    o3.P = p; p += (m + 1 + S) * S;
    o3.C1 = p; p += m * S;
    o3.Cc = p; p += m * S;
    o3.cf1 = p; p += m * S;
    o3.cf2 = p; p += m * S;
    o3.R1 = p; p += S * S;
    o3.R1i = p; p += S * S;
    o3.ri1 = p; p += S * S;
    o3.ri2 = p; p += S * S;
    o3.th = p; p += S;
    o3.res = p; p += S;
    FEDD_TRY(c->d_small.ensure((size_t)p + 8));
    double* Sx = c->d_small.p;
    double* V = c->d_V.p;
    double* xt = c->d_w.p;        // trial solution (ghost tail behind it)
    double* axt = c->d_w.p + nc;  // A xt
    double* z = c->d_Z.p;         // M^-1 v
    double* r = c->d_Z.p + nc;    // residual / V y
    int32_t* d_sa = c->d_flags.p + 12;   // (0 max scratch, 1 bad pivot, 2 DGKS gate, 3-5 coarse setup, 8-11 Schwarz setup)
    const dim3 gn((unsigned)((n + 255) / 256)), blk(256);
    hipStream_t st = c->stream;
    const double chol_tol = c->gmres_chol_tol;
    // k_ss_pass2 keeps the S new Hessenberg columns in dynamic LDS (beside 18 KB of static block matrices): long restart cycles
    // go beyond the 64 KB a kernel gets without asking
    const size_t pass2_lds = (size_t)(S * (m + 2) + 2 * m) * sizeof(double);
    FEDD_CHECK(pass2_lds <= 128 * 1024, "gmres (s-step): restart length %d with blocks of %d vectors exceeds the LDS of the block kernel", m, S);
    if (pass2_lds > 40 * 1024)
        FEDD_HIP(hipFuncSetAttribute((const void*)k_ss_pass2<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pass2_lds));

    auto norm2_into = [&](const double* v, double* out) -> int {
        hipLaunchKernelGGL(k_multidot, dim3(nblk, 1), blk, 0, st, v, ldv, n, 0, v, c->d_part.p, nblk, (const int32_t*)nullptr);
        hipLaunchKernelGGL(k_reduce_cols, dim3(1), blk, 0, st, (const double*)c->d_part.p, out, nblk, (const int32_t*)nullptr);
        return allreduce_sum(c, out, 1);
    };
    // c->gm_mask != nullptr: the constrained system of the GDSW extension solves, A^ = D A D + (I - D), M^^-1 = D M^-1 D + (I - D)
    // (see gmres_solve_dcgs2); monomial blocks only (the shift would have to reach the held rows too)
    const double* mk = c->gm_mask;
    // out = A M^-1 in - theta in  (basis columns: no ghost tail; the shift rides in the SpMV kernel's store)
    auto apply_B = [&](const double* in, double* out, double theta) -> int {
        if (nr > 1) {       // masks ride in the kernels' stores
            double* src = const_cast<double*>(in);
            if (ghosts) {
                src = c->d_Z.p + 2 * nc;
                FEDD_HIP(hipMemcpyAsync(src, in, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
            }
            FEDD_TRY(schwarz_apply_multi(c, src, z, mk));
            return spmm_owned(c, z, out, mk, z);
        }
        if (use_prec) FEDD_TRY(schwarz_apply(c, in, z, false));
        if (mk) {
            if (use_prec) hipLaunchKernelGGL(k_mask_mix, gn, blk, 0, st, mk, (const double*)z, in, z, n);
            else FEDD_HIP(hipMemcpyAsync(z, in, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
            FEDD_TRY(spmv_owned(c, z, out, true));
            hipLaunchKernelGGL(k_mask_mix, gn, blk, 0, st, mk, (const double*)out, (const double*)z, out, n);
            return 0;
        }
        return spmv_owned(c, use_prec ? z : in, out, use_prec, theta != 0.0 ? in : nullptr, theta);
    };

    FEDD_HIP(hipMemsetAsync(Sx + o3.th, 0, (size_t)S * sizeof(double), st));   // monomial block basis (shifts 0)
    if (c->gm_x0 && !mk && nr == 1) {   // "Zero Initial Guess" = false (LinearSolver_def.hpp:76-78): d_x holds x_0, r_0 = b - A x_0
        FEDD_TRY(spmv_owned(c, d_x, axt));
        hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, d_b, -1.0, (const double*)axt, r, n);
    } else {
        FEDD_HIP(hipMemsetAsync(d_x, 0, (size_t)n * sizeof(double), st));  // "Zero Initial Guess" (LinearSolver_def.hpp:76-78)
        FEDD_HIP(hipMemcpyAsync(r, d_b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    FEDD_TRY(norm2_into(r, Sx + o.nrm + 3));
    FEDD_HIP(hipMemcpyAsync(c->h_pinned, Sx + o.nrm + 3, sizeof(double), hipMemcpyDeviceToHost, st));
    FEDD_HIP(hipStreamSynchronize(st));
    const double beta0 = std::sqrt(c->h_pinned[0]);
    int its = 0;
    double relres = beta0 > 0 ? 1.0 : 0.0;
    if (!(beta0 > 0)) {
        if (its_out) *its_out = 0;
        if (relres_out) *relres_out = 0.0;
        return 0;
    }
    hipEvent_t ev = nullptr;
    FEDD_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    struct EvGuard {
        hipEvent_t e;
        ~EvGuard() { (void)hipEventDestroy(e); }
    } ev_guard{ev};
    // column groups in flight per row block of the dot kernel: one when the vectors are long (every workgroup then
    // reads its rows of W once), more to fill the GPU when they are short
    // (round 4, measured at 145 iterations: 1.26 M rows 1 / 2 / 3 / 4 groups 14.60 / 14.74 / 14.99 / 15.10 ms per solve, 1.03 M rows
    // 15.54 / 15.64 / 15.77 / 15.97, 275 k rows 9.24 / 9.32 / 9.29 / 9.09; the old rule, 2048 workgroups in flight, took 2, 3 and 8)
    const int gy_dot = c->gmres_dot_gy > 0 ? c->gmres_dot_gy
                                           : (int)std::max<int64_t>(1, std::min<int64_t>(4, (1024 + nblkd / 2) / std::max(nblkd, 1)));
    double* hout = c->h_pinned + 16;                                  // host mirror of the block result
    double* hout_dev = c->h_pinned_dev ? c->h_pinned_dev + 16 : nullptr;

    // xt = x + M^-1 V(:, 0:cols) y, r = b - A xt, ||r||^2 to the host: the true residual with `cols` columns of this cycle
    // columns whose combination sum_c y_c V_c is already in r: written by the second update of a block that filled its restart
    // cycle (k_blockaxpy<S, true>), -1: none
    int precomb_cols = -1;
    auto trial = [&](int cols, double* true_abs) -> int {
        if (cols > 0) {
            if (cols != precomb_cols) {
                hipLaunchKernelGGL(k_backsolve, dim3(1), dim3(256), (size_t)(cols + 1) * sizeof(double), st, Sx, o, cols, m);
                hipLaunchKernelGGL(k_combine, dim3((unsigned)((n + AX_ROWS - 1) / AX_ROWS)), blk, 0, st, (const double*)V, ldv, n, cols, (const double*)(Sx + o.y), r);
            }
            precomb_cols = -1;      // (r becomes the residual below)
            if (nr > 1) {
                FEDD_TRY(schwarz_apply_multi(c, r, z, mk));
                hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, (const double*)d_x, 1.0, (const double*)z, xt, n);
            } else if (use_prec) {
                FEDD_TRY(schwarz_apply(c, r, z, true));
                if (mk) hipLaunchKernelGGL(k_mask_mix, gn, blk, 0, st, mk, (const double*)z, (const double*)r, z, n);
                hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, (const double*)d_x, 1.0, (const double*)z, xt, n);
            } else {
                hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, (const double*)d_x, 1.0, (const double*)r, xt, n);
            }
        } else {
            FEDD_HIP(hipMemcpyAsync(xt, d_x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        if (nr > 1) {
            FEDD_TRY(spmm_owned(c, xt, axt, mk, xt));
        } else {
            // the residual that decides is formed with the parity CSR (every stored entry), like fedd_spmv, not with the
            // compacted stream the Krylov process runs on (which leaves out sub-ulp cancellation noise)
            FEDD_TRY(spmv_owned(c, xt, axt, true, nullptr, 0.0, 0));
            if (mk) hipLaunchKernelGGL(k_mask_mix, gn, blk, 0, st, mk, (const double*)axt, (const double*)xt, axt, n);
        }
        hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, d_b, -1.0, (const double*)axt, r, n);
        FEDD_TRY(norm2_into(r, Sx + o.nrm + 3));
        FEDD_HIP(hipMemcpyAsync(c->h_pinned, Sx + o.nrm + 3, sizeof(double), hipMemcpyDeviceToHost, st));
        FEDD_HIP(hipStreamSynchronize(st));
        *true_abs = std::sqrt(std::max(c->h_pinned[0], 0.0));
        return 0;
    };
    auto commit = [&]() -> int {   // x <- xt (r and ||r||^2 already belong to it)
        FEDD_HIP(hipMemcpyAsync(d_x, xt, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        return 0;
    };

    // the recurrence residual of a block basis of condition kappa is good to about eps * kappa; measured on the Laplace cubes
    // (monomial basis): it parts from the true residual near 1e-11 for s = 8 and near 2e-13 for s = 4.  Below those floors a
    // claim fails its check and costs a restart, so tight tolerances take shorter blocks from the start.
    // (option "gmres_tol_blocks" 0 lifts the cap: runs that are held to an iteration count instead of a tolerance)
    const int s_tol = !c->gmres_tol_blocks || rtol >= 1e-9 ? 16 : (rtol >= 1e-11 ? 5 : 3);
    const int s_goal = std::max(1, std::min(std::min(c->gmres_s, S), s_tol));
    // Blocks longer than 8 need a better conditioned block basis than the monomial one (its condition grows tenfold every
    // two vectors, 1e7 at s = 8): the Newton basis w_i = (B - theta_i) w_{i-1} with the Ritz values of the first s_goal
    // Arnoldi steps as shifts, Leja-ordered (Bai, Hu, Reichel, "A Newton basis GMRES implementation", 1994).  Until those
    // steps exist the blocks are monomial and at most 8 long.
    const bool newton = c->gmres_newton && s_goal > 8 && !mk && nr == 1;
    bool have_shifts = false;
    std::vector<double> theta((size_t)S, 0.0);
    int s_cur = std::min(s_goal, 8);
    const bool dbg = getenv("FEDD_GMRES_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[gmres s-step] basis at %p (%zu MB), ldv %lld\n", (void*)V, c->d_V.cap * sizeof(double) >> 20, (long long)ldv);
    double tol_abs = rtol * beta0;      // target of the recurrence residual; tightened when the true residual lags behind it
    double last_true = beta0;
    int floor_restarts = 0;             // restarts taken because the true residual stopped following the recurrence
    int nfail = 0;                      // claims of the recurrence that the true residual did not confirm
    double best_fail = 1e300;           // ... and the smallest true residual among them
    int stalls = 0;
    bool done = false;
    c->gmres_blocks = 0;
    c->gmres_cut_blocks = 0;
    c->gmres_fused_blocks = 0;
    while (!done && its < max_it) {
        hipLaunchKernelGGL(k_ss_cycle_init, dim3(1), dim3(256), 0, st, Sx, o, o3, m, S, (const double*)(Sx + o.nrm + 3));
        hipLaunchKernelGGL(k_scale_to, gn, blk, 0, st, (const double*)r, (const double*)(Sx + o.misc + 1), V, n);
        int k = 1;                 // final basis vectors; k - 1 Hessenberg columns are final
        bool restart_now = false;
        int spec_done = 0, spec_k = -1;   // operator applications of the block starting at spec_k that are already in the stream
        while (!done && !restart_now) {
            const int room = std::min(m - (k - 1), max_it - its - (k - 1));
            if (room <= 0) break;
            const int sa = std::min(s_cur, room);
            const int i0 = spec_k == k ? std::min(spec_done, sa) : 0;
            spec_done = 0;
            spec_k = -1;
            for (int i = i0; i < sa; ++i)
                FEDD_TRY(apply_B(V + (int64_t)(k - 1 + i) * ldv, V + (int64_t)(k + i) * ldv, have_shifts ? theta[(size_t)i] : 0.0));
            {
                ScopedTimer t(c, FEDD_T_ORTHO);
                const int ncg = (k + sa + dot_cg - 1) / dot_cg;
                const dim3 gd(nblkd, std::min(gy_dot, ncg));
                auto launch_dot = [&](const dim3& g, const int32_t* dsa) {
                    constexpr int CG0 = ss_cg(S), CG2 = (64 / S < 2 * CG0) ? 64 / S : 2 * CG0;
                    if constexpr (S == 16) {
                        if (dot_split) {
                            hipLaunchKernelGGL((k_blockdot<S, 2, 4, 2>), g, blk, 0, st, (const double*)V, ldv, n, k, sa, dsa, c->d_part.p, nblkd);
                            return;
                        }
                    }
                    if (dotv == 0 || dotv == 4)
                        hipLaunchKernelGGL((k_blockdot<S, 2, CG0>), g, blk, 0, st, (const double*)V, ldv, n, k, sa, dsa, c->d_part.p, nblkd);
                    else if (dotv == 1)
                        hipLaunchKernelGGL((k_blockdot<S, 1, CG0>), g, blk, 0, st, (const double*)V, ldv, n, k, sa, dsa, c->d_part.p, nblkd);
                    else
                        hipLaunchKernelGGL((k_blockdot<S, 1, CG2>), g, blk, 0, st, (const double*)V, ldv, n, k, sa, dsa, c->d_part.p, nblkd);
                };
                const double dot_bytes = 8.0 * (double)n * (k + 2 * sa), upd_bytes = 8.0 * (double)n * (k + 2 * sa);
                {
                    ScopedTimer td(c, FEDD_T_GS_DOT);
                    td.bytes(dot_bytes);
                    launch_dot(gd, (const int32_t*)nullptr);
                }
                hipLaunchKernelGGL(k_reduce_cols, dim3((k + sa) * S), blk, 0, st, (const double*)c->d_part.p, Sx + o3.P, nblkd,
                                   (const int32_t*)nullptr);
                FEDD_TRY(allreduce_sum(c, Sx + o3.P, (k + sa) * S));
                hipLaunchKernelGGL(k_ss_pass1<S>, dim3(1), blk, 0, st, Sx, o3, k, sa, chol_tol, d_sa);
                bool fused = false;
                // (auto: long vectors only -- at 9.9 M rows the step gains 1.1 ms, at the 1.26 M rows of the N = 8 share the two
                //  separate kernels are 0.15 ms ahead: their second read finds much of the basis in the Infinity Cache anyway)
                if constexpr (S == 16) fused = c->gmres_fuse > 0 || (c->gmres_fuse < 0 && n >= 4000000);
                if (fused) {
                    ++c->gmres_fused_blocks;
                    // first update + second dot in one sweep (k_blockfuse), its own class; algorithmic bytes: what the two operations
                    // together must move once -- the basis and the block read, the block written: those of the update alone
                    if constexpr (S == 16) {
                        ScopedTimer tu(c, FEDD_T_GS_FUSED);
                        tu.bytes(upd_bytes);
                        hipLaunchKernelGGL(k_blockfuse<S>, dim3(nblk2), blk, 0, st, V, ldv, n, k, sa, (const int32_t*)d_sa,
                                           (const double*)(Sx + o3.cf1), (const double*)(Sx + o3.ri1), c->d_part.p, nblk2);
                    }
                } else {
                    {
                        ScopedTimer tu(c, FEDD_T_GS_UPDATE);
                        tu.bytes(upd_bytes);
                        hipLaunchKernelGGL(k_blockaxpy<S>, dim3(nblk2), blk, 0, st, V, ldv, n, k, sa, (const int32_t*)d_sa,
                                           (const double*)(Sx + o3.cf1), (const double*)(Sx + o3.ri1));
                    }
                    {
                        ScopedTimer td(c, FEDD_T_GS_DOT);
                        td.bytes(dot_bytes);
                        launch_dot(gd, (const int32_t*)d_sa);
                    }
                }
                hipLaunchKernelGGL(k_reduce_cols, dim3((k + sa) * S), blk, 0, st, (const double*)c->d_part.p, Sx + o3.P, fused ? nblk2 : nblkd,
                                   (const int32_t*)nullptr);
                FEDD_TRY(allreduce_sum(c, Sx + o3.P, (k + sa) * S));
                hipLaunchKernelGGL(k_ss_pass2<S>, dim3(1), blk, pass2_lds, st, Sx, o, o3, k, sa, m,
                                   chol_tol, d_sa, hout_dev);
                if (!hout_dev)
                    FEDD_HIP(hipMemcpyAsync(hout, Sx + o.misc + 4, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
                if (!hout_dev)
                    FEDD_HIP(hipMemcpyAsync(hout + 2, Sx + o3.res, S * sizeof(double), hipMemcpyDeviceToHost, st));
                FEDD_HIP(hipEventRecord(ev, st));
                // a block that fills the restart cycle is followed by the solution update over all its columns: the second
                // update forms that combination while it has the basis in hand (one read of the basis less per cycle)
                bool comb = false;
                if constexpr (S == 16) comb = c->gmres_fuse != 0 && nr == 1 && k - 1 + sa == m && sa >= 2;
                precomb_cols = -1;
                if (comb) {
                    if constexpr (S == 16) {
                        hipLaunchKernelGGL(k_backsolve, dim3(1), dim3(256), (size_t)(k + sa) * sizeof(double), st, Sx, o, k - 1 + sa, m);
                        ScopedTimer tu(c, FEDD_T_GS_UPDATE);
                        tu.bytes(upd_bytes + 8.0 * (double)n);
                        hipLaunchKernelGGL((k_blockaxpy<S, true>), dim3(nblk2), blk, 0, st, V, ldv, n, k, sa, (const int32_t*)d_sa,
                                           (const double*)(Sx + o3.cf2), (const double*)(Sx + o3.ri2), (const double*)(Sx + o.y), r);
                        precomb_cols = k - 1 + sa;
                    }
                } else {
                    ScopedTimer tu(c, FEDD_T_GS_UPDATE);
                    tu.bytes(upd_bytes);
                    hipLaunchKernelGGL(k_blockaxpy<S>, dim3(nblk2), blk, 0, st, V, ldv, n, k, sa, (const int32_t*)d_sa,
                                       (const double*)(Sx + o3.cf2), (const double*)(Sx + o3.ri2));
                }
                t.stop();
            }
            // The head of the next block goes into the stream before the host waits for this block's outcome: the GPU works
            // through the round trip.  It assumes the ordinary outcome (no cut, no convergence, same shifts); otherwise the
            // two applications wrote columns nobody reads.
            if (c->gmres_spec > 0 && !(newton && !have_shifts)) {
                const int k_next = k + sa;
                const int room_next = std::min(m - (k_next - 1), max_it - its - (k_next - 1));
                const int ns = std::min(std::min(c->gmres_spec, s_cur), room_next);
                for (int i = 0; i < ns; ++i)
                    FEDD_TRY(apply_B(V + (int64_t)(k_next - 1 + i) * ldv, V + (int64_t)(k_next + i) * ldv, have_shifts ? theta[(size_t)i] : 0.0));
                if (ns > 0) {
                    spec_done = ns;
                    spec_k = k_next;
                }
            }
            FEDD_HIP(hipEventSynchronize(ev));
            const int sa_eff = (int)hout[0];
            const bool brk = hout[1] != 0.0;
            const int ncolH = brk ? 1 : sa_eff;
            ++c->gmres_blocks;
            if (sa_eff < sa) ++c->gmres_cut_blocks;
            if (sa_eff < sa || brk) precomb_cols = -1;      // (the columns behind a cut are not what y was solved for)
            FEDD_CHECK(ncolH >= 1 && ncolH <= S, "gmres (s-step): block result %d", ncolH);
            for (int i = 0; i < ncolH && !done && !restart_now; ++i) {
                relres = hout[2 + i] / beta0;
                if (hout[2 + i] <= tol_abs) {
                    const int cols = k - 1 + i + 1;
                    double ta = 0.0;
                    FEDD_TRY(trial(cols, &ta));
                    if (dbg) fprintf(stderr, "[gmres s-step] k %d block col %d: claim with %d columns, recurrence %.3e true %.3e target %.3e (s %d)\n",
                                     k, i, cols, hout[2 + i] / beta0, ta / beta0, rtol, s_cur);
                    if (ta <= rtol * beta0) {
                        FEDD_TRY(commit());
                        its += cols;
                        relres = ta / beta0;
                        done = true;
                    } else if (nfail >= 1 && ta >= 0.5 * best_fail && ta <= 100.0 * rtol * beta0 && floor_restarts < 1) {
                        // the true residual has stopped following the recurrence.  That is either the rounding floor of
                        // b - A x or only the floor of THIS cycle's recurrence (a cycle is good to about eps * kappa(block
                        // basis) relative to the residual it started from): one restart from the true residual tells them
                        // apart -- a new cycle reaches the target if it is reachable (the effect of iterative refinement)
                        ++floor_restarts;
                        if (ta < last_true) {
                            FEDD_TRY(commit());
                            last_true = ta;
                        } else {
                            FEDD_TRY(trial(0, &ta));
                        }
                        its += cols;
                        relres = last_true / beta0;
                        nfail = 0;
                        best_fail = 1e300;
                        tol_abs = rtol * beta0;
                        restart_now = true;
                    } else if (nfail >= 1 && ta >= 0.5 * best_fail && ta <= 100.0 * rtol * beta0) {
                        // the recurrence keeps falling, the true residual of the computed x does not follow any more: b - A x
                        // has reached its rounding floor (badly scaled systems, tolerances near 1e-13).  The one-vector
                        // solvers stop on the recurrence alone; here the iteration ends once the floor is evident.  What is
                        // reported is the TRUE residual (it may exceed rtol by up to 100x: the caller sees that);
                        // fedd_gmres_status tells that the floor was reached and gives the recurrence residual beside it.
                        FEDD_TRY(commit());
                        its += cols;
                        relres = ta / beta0;
                        c->gmres_floor = 1;
                        c->gmres_rec_relres = hout[2 + i] / beta0;
                        done = true;
                    } else if (ta > 4.0 * std::max(hout[2 + i], 1e-300) || !(ta < last_true)) {
                        // the recurrence has lost touch with the true residual: keep what was gained, restart
                        ++nfail;
                        best_fail = std::min(best_fail, ta);
                        if (ta < last_true) {
                            FEDD_TRY(commit());
                            last_true = ta;
                            its += cols;
                        } else {
                            FEDD_TRY(trial(0, &ta));   // r, ||r||^2 of the current x again
                            its += cols;
                            ++stalls;
                        }
                        relres = last_true / beta0;
                        s_cur = std::max(1, s_cur / 2);
                        restart_now = true;
                    } else {
                        ++nfail;
                        best_fail = std::min(best_fail, ta);
                        tol_abs *= 0.9 * rtol * beta0 / ta;
                    }
                }
            }
            if (done || restart_now) break;
            if (brk) {   // the Krylov space is exhausted (or numerically so): the true residual decides
                double ta = 0.0;
                FEDD_TRY(trial(k, &ta));
                its += k;
                if (ta < last_true) {
                    FEDD_TRY(commit());
                    last_true = ta;
                } else {
                    FEDD_TRY(trial(0, &ta));
                    ++stalls;
                }
                relres = last_true / beta0;
                if (last_true <= rtol * beta0) done = true;
                restart_now = true;
                break;
            }
            k += sa_eff;
            if (sa_eff < sa) s_cur = std::max(1, sa_eff);   // the block basis became dependent: shorter blocks from here on
            else if (newton && !have_shifts && k - 1 >= s_goal) {
                // the shifts, once per solve: Ritz values of the leading s_goal x s_goal Hessenberg matrix
                std::vector<double> Hh((size_t)s_goal * s_goal);
                FEDD_HIP(hipMemcpy2DAsync(Hh.data(), (size_t)s_goal * sizeof(double), Sx + o3.Hraw, (size_t)(m + 1) * sizeof(double),
                                          (size_t)s_goal * sizeof(double), (size_t)s_goal, hipMemcpyDeviceToHost, st));
                FEDD_HIP(hipStreamSynchronize(st));
                hessenberg_real_parts(Hh, s_goal, s_goal, theta.data());
                leja_order(theta.data(), s_goal);
                bool finite = true;
                for (int i = 0; i < s_goal; ++i) finite = finite && std::isfinite(theta[(size_t)i]);
                if (finite) {
                    for (int i = 0; i < S; ++i) c->h_pinned[64 + i] = i < s_goal ? theta[(size_t)i] : 0.0;
                    FEDD_HIP(hipMemcpyAsync(Sx + o3.th, c->h_pinned + 64, (size_t)S * sizeof(double), hipMemcpyHostToDevice, st));
                    have_shifts = true;
                    s_cur = s_goal;
                    if (dbg) {
                        fprintf(stderr, "[gmres s-step] Newton shifts:");
                        for (int i = 0; i < s_goal; ++i) fprintf(stderr, " %.4f", theta[(size_t)i]);
                        fprintf(stderr, "\n");
                    }
                }
            }
        }
        if (done) break;
        if (!restart_now) {   // cycle used up (restart length or iteration limit)
            double ta = 0.0;
            FEDD_TRY(trial(k - 1, &ta));
            its += k - 1;
            if (ta < 0.999 * last_true) stalls = 0;
            else ++stalls;
            if (ta < last_true) {
                FEDD_TRY(commit());
                last_true = ta;
            } else {
                FEDD_TRY(trial(0, &ta));
            }
            relres = last_true / beta0;
            if (last_true <= rtol * beta0) done = true;
        }
        if (stalls >= 3) {        // no progress in three cycles: give the caller the residual reached
            c->gmres_floor = 2;
            break;
        }
    }
    FEDD_HIP(hipGetLastError());
    FEDD_HIP(hipStreamSynchronize(st));
    if (its_out) *its_out = its;
    if (relres_out) *relres_out = relres;
    return 0;
}

int gmres_solve(fedd_ctx* c, const double* d_b, double* d_x, double rtol, int max_it, int restart, int use_prec,
                int* its_out, double* relres_out) {
    c->gmres_floor = 0;
    c->gmres_rec_relres = -1.0;
    if (c->gmres_kind == 2) {
        // block length: "gmres_s" 0 = by the vector length per rank -- 16-vector (Newton-basis) blocks where the sweeps over the
        // basis dominate, 8-vector blocks on short vectors, where the longer blocks' fixed costs (two monomial blocks first,
        // a 16 x 16 Cholesky per pass, one more vector per SpMV) are not paid back.  Measured: 1.03 M rows / 71 iterations (cfg 2)
        // 11.8 ms with 8, 12.2 with 16; 1.26 M rows / 145 iterations (one GPU's share of cfg 3) 24.4 against 22.5; 9.9 M rows
        // 122.5 against 108.0.  Every rank must take the same one: the decision uses the global row count
        if (c->gmres_s == 0) {
            double ng = (double)c->n_rows;
            if (c->nranks > 1) {
                FEDD_TRY(c->d_small.ensure(std::max<size_t>(16, c->d_small.cap)));
                FEDD_HIP(hipMemcpyAsync(c->d_small.p, &ng, sizeof(double), hipMemcpyHostToDevice, c->stream));
                FEDD_HIP(hipStreamSynchronize(c->stream));
                FEDD_TRY(allreduce_sum(c, c->d_small.p, 1));
                FEDD_HIP(hipMemcpyAsync(&ng, c->d_small.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
                FEDD_HIP(hipStreamSynchronize(c->stream));
            }
            c->gmres_s = ng / c->nranks >= 1.2e6 ? 16 : 8;
            const int rc = gmres_solve(c, d_b, d_x, rtol, max_it, restart, use_prec, its_out, relres_out);
            c->gmres_s_used = c->gmres_s;
            c->gmres_s = 0;
            return rc;
        }
        c->gmres_s_used = c->gmres_s;
        // (the constrained solves run monomial blocks of at most eight vectors: the 8-column kernels)
        if (c->gm_mask && c->gmres_s > 8) return gmres_solve_sstep<8>(c, d_b, d_x, rtol, max_it, restart, use_prec, its_out, relres_out);
        // (cycles of more than 800 vectors: the 16-column block kernel's LDS image of its new Hessenberg columns does not fit)
        if (c->gmres_s > 8 && std::min(restart, max_it) > 800)
            return gmres_solve_sstep<8>(c, d_b, d_x, rtol, max_it, restart, use_prec, its_out, relres_out);
        if (c->gmres_s <= 4) return gmres_solve_sstep<4>(c, d_b, d_x, rtol, max_it, restart, use_prec, its_out, relres_out);
        if (c->gmres_s <= 8) return gmres_solve_sstep<8>(c, d_b, d_x, rtol, max_it, restart, use_prec, its_out, relres_out);
        return gmres_solve_sstep<16>(c, d_b, d_x, rtol, max_it, restart, use_prec, its_out, relres_out);
    }
    if (c->gmres_kind == 0 || c->gmres_kind == 2) return gmres_solve_dcgs2(c, d_b, d_x, rtol, max_it, restart, use_prec, its_out, relres_out);
    const int64_t n = c->n_rows;
    const int m = std::min(restart, max_it);
    const int64_t ldv = (n + 15) & ~(int64_t)15;  // 128-byte aligned basis columns
    FEDD_CHECK(m + 2 <= 1024, "gmres: restart length above 1022 is not supported");
    const int nblk = (int)((n + MD_ROWS - 1) / MD_ROWS), nblk2 = (int)((n + AX_ROWS - 1) / AX_ROWS);
    FEDD_TRY(c->d_V.ensure((size_t)(m + 1) * ldv));
    if (c->gm_V_ldv != ldv) c->gm_V_ldv = -1;
    FEDD_TRY(c->d_w.ensure(std::max<size_t>((size_t)n, c->d_w.cap)));
    FEDD_TRY(c->d_Z.ensure((size_t)n * 2));
    FEDD_TRY(c->d_part.ensure(std::max((size_t)(m + 2) * nblk, (size_t)nblk2)));
    Off o;
    int p = 0;
    o.H = p; p += (m + 1) * m;
    o.cs = p; p += m;
    o.sn = p; p += m;
    o.g = p; p += m + 1;
    o.h1 = p; p += m + 2;
    o.h2 = p; p += m + 2;
    o.nrm = p; p += 4;
    o.y = p; p += m;
    o.misc = p; p += 8;
    FEDD_TRY(c->d_small.ensure((size_t)p + 8));
    double* S = c->d_small.p;
    FEDD_TRY(c->d_flags.ensure(16));
    int32_t* gate = c->d_flags.p + 2;
    double* V = c->d_V.p;
    double* w = c->d_w.p;
    double* z = c->d_Z.p;       // M^-1 v
    double* r = c->d_Z.p + n;   // residual / u
    const dim3 gn((unsigned)((n + 255) / 256)), blk(256);
    hipStream_t st = c->stream;

    auto norm2_into = [&](const double* v, double* out) -> int {  // out[0] = v.v (global)
        hipLaunchKernelGGL(k_multidot, dim3(nblk, 1), blk, 0, st, v, ldv, n, 0, v, c->d_part.p, nblk, (const int32_t*)nullptr);
        hipLaunchKernelGGL(k_reduce_cols, dim3(1), blk, 0, st, (const double*)c->d_part.p, out, nblk, (const int32_t*)nullptr);
        return allreduce_sum(c, out, 1);
    };
    auto residual = [&]() -> int {  // r = b - A x
        FEDD_TRY(spmv_owned(c, d_x, r));
        hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, d_b, -1.0, (const double*)r, r, n);
        return 0;
    };

    if (c->gm_x0) {   // "Zero Initial Guess" = false (LinearSolver_def.hpp:76-78): d_x holds x_0, r_0 = b - A x_0
        FEDD_TRY(residual());
    } else {
        FEDD_HIP(hipMemsetAsync(d_x, 0, (size_t)n * sizeof(double), st));  // "Zero Initial Guess" (LinearSolver_def.hpp:76-78)
        FEDD_HIP(hipMemcpyAsync(r, d_b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    FEDD_TRY(norm2_into(r, S + o.nrm + 3));
    FEDD_HIP(hipMemcpyAsync(c->h_pinned, S + o.nrm + 3, sizeof(double), hipMemcpyDeviceToHost, st));
    FEDD_HIP(hipStreamSynchronize(st));
    const double beta0 = std::sqrt(c->h_pinned[0]);
    int its = 0;
    double relres = beta0 > 0 ? 1.0 : 0.0;
    if (!(beta0 > 0)) {
        if (its_out) *its_out = 0;
        if (relres_out) *relres_out = 0.0;
        return 0;
    }
    // The convergence test of iteration j is read on the host while iteration j + 1 is already
    // queued (two pinned slots, one event each): the device never waits for the host round trip.
    // An iteration queued past convergence only writes basis column j + 2, Hessenberg column
    // j + 1 and g[j+1..j+2], none of which the update of x with k = j + 1 columns reads.
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int q = 0; q < 2; ++q) FEDD_HIP(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));
    struct EvGuard {
        hipEvent_t* e;
        ~EvGuard() {
            for (int q = 0; q < 2; ++q)
                if (e[q]) (void)hipEventDestroy(e[q]);
        }
    } ev_guard{ev};
    bool converged = false;
    while (!converged && its < max_it) {
        hipLaunchKernelGGL(k_cycle_init, dim3(1), dim3(1), 0, st, S, o, m, (const double*)(S + o.nrm + 3));
        hipLaunchKernelGGL(k_scale_to, gn, blk, 0, st, (const double*)r, (const double*)(S + o.misc + 1), V, n);
        int k = 0;
        int issued = its, checked = 0, queued = 0;  // iterations of this cycle: results read / queued
        auto check = [&](int jj) -> int {           // 1 = stop (converged or breakdown), < 0 = error
            if (hipEventSynchronize(ev[jj & 1]) != hipSuccess) return -1;
            const double* hp = c->h_pinned + 4 * (jj & 1);
            ++its;
            ++checked;
            k = jj + 1;
            relres = hp[0] / beta0;
            const bool breakdown = !(hp[2] > 0.0);   // ||w|| after both passes, behind the DGKS gate: a true breakdown
            return (relres <= rtol || breakdown) ? 1 : 0;
        };
        for (int j = 0; j < m && issued < max_it; ++j) {
            const double* vj = V + (int64_t)j * ldv;
            if (use_prec) FEDD_TRY(schwarz_apply(c, vj, z));
            FEDD_TRY(spmv_owned(c, use_prec ? z : vj, w));
            {
                ScopedTimer t(c, FEDD_T_ORTHO);
                // pass 1: h1 = V^T w, nrm[0] = w.w ; w -= V h1, nrm[1] = ||w||^2
                hipLaunchKernelGGL(k_multidot, dim3(nblk, (j + 2 + MD_CG - 1) / MD_CG), blk, 0, st, (const double*)V, ldv, n,
                                   j + 1, (const double*)w, c->d_part.p, nblk, (const int32_t*)nullptr);
                hipLaunchKernelGGL(k_reduce_cols, dim3(j + 2), blk, 0, st, (const double*)c->d_part.p, S + o.h1, nblk,
                                   (const int32_t*)nullptr);
                FEDD_TRY(allreduce_sum(c, S + o.h1, j + 2));
                // h1[j+1] now holds w.w: gate and the norm after pass 1 follow from h1 alone
                hipLaunchKernelGGL(k_dgks_gate, dim3(1), dim3(1), 0, st, (const double*)(S + o.h1), j, gate, S + o.nrm + 1);
                hipLaunchKernelGGL(k_multiaxpy, dim3(nblk2), blk, 0, st, (const double*)V, ldv, n, j + 1,
                                   (const double*)(S + o.h1), w, c->d_part.p, (const int32_t*)nullptr);
                // pass 2 (gated on the device; on several ranks the gate is identical everywhere
                // because it is computed from all-reduced numbers, so the collectives stay matched)
                hipLaunchKernelGGL(k_multidot, dim3(nblk, (j + 2 + MD_CG - 1) / MD_CG), blk, 0, st, (const double*)V, ldv, n,
                                   j + 1, (const double*)w, c->d_part.p, nblk, (const int32_t*)gate);
                hipLaunchKernelGGL(k_reduce_cols, dim3(j + 1), blk, 0, st, (const double*)c->d_part.p, S + o.h2, nblk,
                                   (const int32_t*)gate);
                FEDD_TRY(allreduce_sum(c, S + o.h2, j + 1));
                hipLaunchKernelGGL(k_multiaxpy, dim3(nblk2), blk, 0, st, (const double*)V, ldv, n, j + 1,
                                   (const double*)(S + o.h2), w, c->d_part.p, (const int32_t*)gate);
                if (c->nranks > 1) {
                    hipLaunchKernelGGL(k_reduce_cols, dim3(1), blk, 0, st, (const double*)c->d_part.p, S + o.nrm + 2, nblk2,
                                       (const int32_t*)gate);
                    FEDD_TRY(allreduce_sum(c, S + o.nrm + 2, 1));
                }
                t.stop();
            }
            hipLaunchKernelGGL(k_givens, dim3(1), dim3(256), (size_t)(3 * m + 2) * sizeof(double), st, S, o, j, m,
                               (const int32_t*)gate, (const double*)(c->nranks > 1 ? nullptr : c->d_part.p), nblk2);
            hipLaunchKernelGGL(k_scale_to, gn, blk, 0, st, (const double*)w, (const double*)(S + o.misc + 1),
                               V + (int64_t)(j + 1) * ldv, n);
            FEDD_HIP(hipMemcpyAsync(c->h_pinned + 4 * (j & 1), S + o.misc, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
            FEDD_HIP(hipEventRecord(ev[j & 1], st));
            ++issued;
            ++queued;
            if (j > 0) {
                const int rc = check(j - 1);
                FEDD_CHECK(rc >= 0, "gmres: waiting for iteration %d failed", j - 1);
                if (rc) {
                    converged = true;
                    break;
                }
            }
        }
        if (!converged && checked < queued) {
            const int rc = check(queued - 1);
            FEDD_CHECK(rc >= 0, "gmres: waiting for iteration %d failed", queued - 1);
            converged = rc != 0;
        }
        // x += M^-1 (V y)
        hipLaunchKernelGGL(k_backsolve, dim3(1), dim3(256), (size_t)(k + 1) * sizeof(double), st, S, o, k, m);
        hipLaunchKernelGGL(k_combine, dim3((unsigned)((n + AX_ROWS - 1) / AX_ROWS)), blk, 0, st, (const double*)V, ldv, n, k, (const double*)(S + o.y), r);
        if (use_prec) {
            FEDD_TRY(schwarz_apply(c, r, z));
            hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, (const double*)d_x, 1.0, (const double*)z, d_x, n);
        } else {
            hipLaunchKernelGGL(k_axpby, gn, blk, 0, st, 1.0, (const double*)d_x, 1.0, (const double*)r, d_x, n);
        }
        if (!converged && its < max_it) {
            FEDD_TRY(residual());
            FEDD_TRY(norm2_into(r, S + o.nrm + 3));
        }
    }
    FEDD_HIP(hipGetLastError());
    FEDD_HIP(hipStreamSynchronize(st));
    if (its_out) *its_out = its;
    if (relres_out) *relres_out = relres;
    return 0;
}

}  // namespace fedd
