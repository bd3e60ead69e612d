// Batched dense inversion on the f64 matrix cores (v_mfma_f64_16x16x4_f64): blocked Gauss-Jordan, block size 64,
// no pivoting.  Used for
//   * the coarse matrix K0 of the second Schwarz level (one SPD matrix, coarse.hip), and
//   * the local matrices of Schwarz subdomains beyond the register-tiled classes (161 .. 1024 dofs, schwarz_big.hip):
//     principal submatrices that are unit rows (Dirichlet) plus an SPD block, or -- merged Stokes systems -- a
//     saddle-point block ordered velocities first, pressures second (the elimination is then a block LDL^T: SPD
//     velocity block, negative definite Schur complement; no row exchanges needed).
// These are the dense, GEMM-shaped contractions of the path (the role KLU / Amesos2 play behind FROSch's
// "Solver" entries, feddlib/problems/tests/laplace/parametersPrec.xml:33-47, stokes/parametersPrec.xml:30-35).
//
// Step kb with pivot block D = K[kb,kb]:  K[kb,kb] <- D^-1, K[kb,j] <- D^-1 K[kb,j],
// K[i,kb] <- -K[i,kb] D^-1, K[i,j] <- K[i,j] - K[i,kb] D^-1 K[kb,j]  (i, j != kb).
// Three launches per step over the whole batch; matrix b occupies K + b * stride (leading dimension ld, a multiple
// of 64; rows / columns beyond the matrix' own size are identity padding) and only its first nblk[b] block rows /
// columns are touched.
#include "fedd_internal.hpp"

namespace fedd {
namespace {

constexpr int NB = 64;

// D^-1 of the 64 x 64 pivot block, register tiled (the scheme of schwarz.hip k_invert_reg, T = 4)
__global__ __launch_bounds__(256) void k_diag_inv(const double* __restrict__ Kall, int64_t ld, int64_t stride, int kb,
                                                  const int32_t* __restrict__ nblk, int spd,
                                                  double* __restrict__ Dall, int32_t* __restrict__ bad) {
    constexpr int T = 4;
    __shared__ double colbuf[2][NB], rowbuf[2][NB];
    const int bz = blockIdx.x;
    if (nblk && kb >= nblk[bz]) return;
    const double* __restrict__ K = Kall + (int64_t)bz * stride;
    double* __restrict__ Dinv = Dall + (int64_t)bz * NB * NB;
    const int tid = threadIdx.x, ty = tid & 15, tx = tid >> 4;
    const double* __restrict__ D = K + ((int64_t)kb * NB) * ld + (int64_t)kb * NB;
    double A[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) A[a][b] = D[(int64_t)(ty + 16 * a) * ld + tx + 16 * b];
    bool singular = false;
#pragma unroll
    for (int kq = 0; kq < T; ++kq) {
#pragma unroll 1
        for (int kc = 0; kc < 16; ++kc) {
            const int k = 16 * kq + kc;
            const int buf = k & 1;
            if (tx == kc) {
#pragma unroll
                for (int a = 0; a < T; ++a) colbuf[buf][ty + 16 * a] = A[a][kq];
            }
            if (ty == kc) {
#pragma unroll
                for (int b = 0; b < T; ++b) rowbuf[buf][tx + 16 * b] = A[kq][b];
            }
            __syncthreads();
            const double piv = rowbuf[buf][k];
            singular = singular || !((spd ? piv : fabs(piv)) > 1e-300);
            const double pinv = 1.0 / piv;
            double cc[T], rr[T];
#pragma unroll
            for (int a = 0; a < T; ++a) cc[a] = colbuf[buf][ty + 16 * a];
#pragma unroll
            for (int b = 0; b < T; ++b) rr[b] = rowbuf[buf][tx + 16 * b] * pinv;
            if (ty == kc) {
                cc[kq] = -1.0;
#pragma unroll
                for (int b = 0; b < T; ++b) A[kq][b] = 0.0;
            }
            if (tx == kc) {
                rr[kq] = pinv;
#pragma unroll
                for (int a = 0; a < T; ++a) A[a][kq] = 0.0;
            }
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = 0; b < T; ++b) A[a][b] = fma(-cc[a], rr[b], A[a][b]);
        }
    }
    if (singular && tid == 0) bad[0] = 1;
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) Dinv[(ty + 16 * a) * NB + tx + 16 * b] = A[a][b];
}

typedef double double4_t __attribute__((ext_vector_type(4)));

// 64 x 64 x 64 product on the f64 matrix cores: the four waves of the workgroup each own a 32 x 32
// quadrant (2 x 2 tiles of v_mfma_f64_16x16x4_f64).  A-fragment lane l = A[l & 15][k = l >> 4],
// B-fragment B[k = l >> 4][l & 15], result register q of lane l = C[(l >> 4) + 4 q][l & 15].
constexpr int LDA_S = 68, LDB_S = 80;  // LDS leading dimensions: the fragment reads are 2-way at worst

__device__ __forceinline__ void mm64(const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                     int64_t ldb, double* As, double* Bs, double4_t acc[2][2]) {
    const int tid = threadIdx.x;
    {
        // all 32 loads of a lane in flight before the first store to LDS (a load -> store loop waited for memory sixteen times
        // per tile: k_update 0.41 ms per step on the 6 591^2 coarse matrix of GDSW)
        constexpr int U = NB * NB / 256;
        double ra[U], rb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = tid + 256 * u, r = e >> 6, cidx = e & 63;
            ra[u] = A[(int64_t)r * lda + cidx];
            rb[u] = B[(int64_t)r * ldb + cidx];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = tid + 256 * u, r = e >> 6, cidx = e & 63;
            As[r * LDA_S + cidx] = ra[u];
            Bs[r * LDB_S + cidx] = rb[u];
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int r0 = 32 * (wave >> 1), c0 = 32 * (wave & 1);
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int s = 0; s < NB / 4; ++s) {
        const double a0 = As[(r0 + li) * LDA_S + 4 * s + lk];
        const double a1 = As[(r0 + 16 + li) * LDA_S + 4 * s + lk];
        const double b0 = Bs[(4 * s + lk) * LDB_S + c0 + li];
        const double b1 = Bs[(4 * s + lk) * LDB_S + c0 + 16 + li];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// element (row, col) of the workgroup's 64 x 64 tile held in acc[ti][tj] register q
#define MM64_FOR_EACH(BODY)                                                     \
    {                                                                           \
        const int wave_ = threadIdx.x >> 6, lane_ = threadIdx.x & 63;           \
        _Pragma("unroll") for (int ti = 0; ti < 2; ++ti)                        \
        _Pragma("unroll") for (int tj = 0; tj < 2; ++tj)                        \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                         \
            const int row = 32 * (wave_ >> 1) + 16 * ti + (lane_ >> 4) + 4 * q; \
            const int col = 32 * (wave_ & 1) + 16 * tj + (lane_ & 15);          \
            const double v = acc[ti][tj][q];                                    \
            BODY                                                                \
        }                                                                       \
    }

// column block bj: save the column panel tile C[bj] = K[bj, kb] and form R[bj] = Dinv K[kb, bj]
__global__ __launch_bounds__(256) void k_panels(const double* __restrict__ Kall, int64_t ld, int64_t stride, int kb,
                                                const int32_t* __restrict__ nblk, const double* __restrict__ Dall,
                                                double* __restrict__ Rall, double* __restrict__ Call) {
    __shared__ double As[NB * LDA_S];
    __shared__ double Bs[NB * LDB_S];
    const int bj = blockIdx.x, bz = blockIdx.y;
    if (nblk && (kb >= nblk[bz] || bj >= nblk[bz])) return;
    const double* __restrict__ K = Kall + (int64_t)bz * stride;
    const double* __restrict__ Dinv = Dall + (int64_t)bz * NB * NB;
    double* __restrict__ R = Rall + (int64_t)bz * NB * ld;
    double* __restrict__ Cp = Call + (int64_t)bz * NB * ld;
    {
        constexpr int U = NB * NB / 256;
        double rc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = threadIdx.x + 256 * u, r = e >> 6, cidx = e & 63;
            rc[u] = K[((int64_t)bj * NB + r) * ld + (int64_t)kb * NB + cidx];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = threadIdx.x + 256 * u, r = e >> 6, cidx = e & 63;
            Cp[((int64_t)bj * NB + r) * NB + cidx] = rc[u];
        }
    }
    if (bj == kb) return;
    double4_t acc[2][2];
    mm64(Dinv, NB, K + ((int64_t)kb * NB) * ld + (int64_t)bj * NB, ld, As, Bs, acc);
    MM64_FOR_EACH(R[(int64_t)row * ld + (int64_t)bj * NB + col] = v;)
}

__global__ __launch_bounds__(256) void k_update(double* __restrict__ Kall, int64_t ld, int64_t stride, int kb,
                                                const int32_t* __restrict__ nblk, const double* __restrict__ Dall,
                                                const double* __restrict__ Rall, const double* __restrict__ Call) {
    __shared__ double As[NB * LDA_S];
    __shared__ double Bs[NB * LDB_S];
    const int bi = blockIdx.y, bj = blockIdx.x, bz = blockIdx.z;
    if (nblk && (kb >= nblk[bz] || bi >= nblk[bz] || bj >= nblk[bz])) return;
    double* __restrict__ K = Kall + (int64_t)bz * stride;
    const double* __restrict__ Dinv = Dall + (int64_t)bz * NB * NB;
    const double* __restrict__ R = Rall + (int64_t)bz * NB * ld;
    const double* __restrict__ Cp = Call + (int64_t)bz * NB * ld;
    double* __restrict__ tile = K + ((int64_t)bi * NB) * ld + (int64_t)bj * NB;
    if (bi == kb) {
        const double* __restrict__ src = bj == kb ? Dinv : R + (int64_t)bj * NB;
        const int64_t lds = bj == kb ? NB : ld;
        constexpr int U = NB * NB / 256;
        double rc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = threadIdx.x + 256 * u, r = e >> 6, cidx = e & 63;
            rc[u] = src[(int64_t)r * lds + cidx];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = threadIdx.x + 256 * u, r = e >> 6, cidx = e & 63;
            tile[(int64_t)r * ld + cidx] = rc[u];
        }
        return;
    }
    double4_t acc[2][2];
    if (bj == kb) {
        mm64(Cp + (int64_t)bi * NB * NB, NB, Dinv, NB, As, Bs, acc);
        MM64_FOR_EACH(tile[(int64_t)row * ld + col] = -v;)
    } else {
        mm64(Cp + (int64_t)bi * NB * NB, NB, R + (int64_t)bj * NB, ld, As, Bs, acc);
        MM64_FOR_EACH(tile[(int64_t)row * ld + col] -= v;)
    }
}

}  // namespace

// In-place inverses of `batch` matrices K + b * stride (leading dimension ld, multiple of 64).  d_nblk (nullable):
// per matrix the number of 64-blocks that are not identity padding.  spd != 0: a pivot <= 0 raises the flag;
// spd == 0: a pivot of magnitude <= 1e-300 does.
int dense_invert_batched(fedd_ctx* c, double* K, int64_t ld, int batch, int64_t stride, const int32_t* d_nblk,
                         int max_nblk, int spd, int32_t* d_bad) {
    FEDD_CHECK(ld % NB == 0 && batch >= 1 && max_nblk >= 1 && max_nblk <= ld / NB, "dense_invert_batched: ld %lld batch %d",
               (long long)ld, batch);
    FEDD_CHECK(batch <= 65535, "dense_invert_batched: at most 65535 matrices per call");
    FEDD_TRY(c->d_dense_ws.ensure((size_t)batch * ((size_t)NB * NB + 2 * (size_t)NB * ld)));
    double* Dinv = c->d_dense_ws.p;
    double* R = Dinv + (size_t)batch * NB * NB;
    double* Cp = R + (size_t)batch * NB * ld;
    for (int kb = 0; kb < max_nblk; ++kb) {
        hipLaunchKernelGGL(k_diag_inv, dim3(batch), dim3(256), 0, c->stream, (const double*)K, ld, stride, kb, d_nblk, spd,
                           Dinv, d_bad);
        hipLaunchKernelGGL(k_panels, dim3(max_nblk, batch), dim3(256), 0, c->stream, (const double*)K, ld, stride, kb, d_nblk,
                           (const double*)Dinv, R, Cp);
        hipLaunchKernelGGL(k_update, dim3(max_nblk, max_nblk, batch), dim3(256), 0, c->stream, K, ld, stride, kb, d_nblk,
                           (const double*)Dinv, (const double*)R, (const double*)Cp);
    }
    FEDD_HIP(hipGetLastError());
    return 0;
}

}  // namespace fedd
