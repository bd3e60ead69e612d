// Sixteen right-hand sides at a time: the operator and the one-level Schwarz preconditioner on "stacked" vectors
//   X[row * 16 + j],  j = right-hand side,
// for the interior extension solves of the GDSW / RGDSW coarse space (coarse.hip, gdsw_setup): all (class, component)
// columns of Phi solve with the same matrix, so the matrix and the local inverses are read once per sweep for sixteen
// columns instead of once per column (FROSch computes the extensions column block by column block with one factorisation,
// FROSch_HarmonicCoarseOperator_def.hpp, reached from feddlib/problems/Solver/Preconditioner_def.hpp:243-330).
//   k_spmm         Y = A X on the assembled CSR: eight lanes per row, a 16-byte gather of X[col][2 jj .. 2 jj + 1] per lane and
//                  entry -- the sixteen values of a column index are one 128-byte line; the entries themselves pass through
//                  LDS.  HBM traffic: 12 B per entry once + 2 * 128 B per row, against sixteen passes over the matrix.
//   k_apply_multi  Z = M^-1 R (restricted additive Schwarz): one wave per subdomain, Z_own[16 RT x 16] = Ainv[16 RT x n] R_sub[n x 16]
//                  on the f64 matrix cores (v_mfma_f64_16x16x4_f64: M = owned rows, N = right-hand sides, K = subdomain dofs);
//                  the N dimension that k_apply_mfma (schwarz.hip) fills with sixteen subdomains is filled with sixteen columns.
// Both take the 0 / 1 mask of the constrained operator (gmres.hip, gm_mask) in their store:  out = mask ? result : alt.
#include "fedd_internal.hpp"

namespace fedd {
namespace {

constexpr int NMAX = SCHWARZ_NMAX;
typedef double mr_d4 __attribute__((ext_vector_type(4)));
typedef double mr_d2 __attribute__((ext_vector_type(2)));

// 32 rows per workgroup; their entries -- contiguous in the CSR arrays -- are staged through LDS in chunks of SPMM_CHUNK
// (coalesced reads of colind / val, once), then lane (row slot, jj) walks the entries of its row in order: one accumulator
// pair per row, so the sums do not depend on where the chunks fall.  Measured at cfg 5's share (2.57 M rows, 113.8 M entries): 0.72 ms
// per sweep of sixteen columns (entries read per lane straight from memory: 1.16 ms; the three rows of a node block sharing their
// gathers of X, entries again per lane: 1.18 ms; the same with the blocks' entries staged through LDS: 1.07 ms -- the gathered
// bytes are not what costs; four or two lanes per row with 32 / 64 bytes gathered per lane and entry: 0.74 / 0.84 ms)
constexpr int SPMM_CHUNK = 2048;
__global__ __launch_bounds__(256) void k_spmm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                              const double* __restrict__ val, int64_t n_rows, const double* __restrict__ X,
                                              double* __restrict__ Y, const double* __restrict__ mk, const double* __restrict__ alt) {
    constexpr int NR = MULTI_NR, LPR = NR / 2, ROWS = 256 / LPR;
    __shared__ double sv[SPMM_CHUNK];
    __shared__ int32_t sc[SPMM_CHUNK];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * ROWS;
    const int64_t row = r0 + tid / LPR;
    const int jj = tid % LPR;
    const bool live = row < n_rows;
    const int32_t wb = rowptr[r0], we = rowptr[min(r0 + ROWS, n_rows)];
    const int32_t b = live ? rowptr[row] : we, e = live ? rowptr[row + 1] : we;
    const mr_d2* __restrict__ X2 = reinterpret_cast<const mr_d2*>(X) + jj;
    mr_d2 acc = {0.0, 0.0};
    for (int32_t cs = wb; cs < we; cs += SPMM_CHUNK) {
        const int32_t ce = min(cs + SPMM_CHUNK, we);
        if (cs != wb) __syncthreads();
        {
            // every load of the chunk in flight before the first store to LDS (a load -> store loop waits for memory once per
            // trip: 0.72 ms per sweep instead of the figure below)
            constexpr int U = SPMM_CHUNK / 256;
            int32_t rc[U];
            double rv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int32_t q = cs + tid + 256 * u;
                rc[u] = q < ce ? colind[q] : 0;
                rv[u] = q < ce ? val[q] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int32_t q = cs + tid + 256 * u;
                if (q < ce) {
                    sc[q - cs] = rc[u];
                    sv[q - cs] = rv[u];
                }
            }
        }
        __syncthreads();
        int32_t p = max(b, cs) - cs;
        const int32_t pe = min(e, ce) - cs;
        for (; p + 3 < pe; p += 4) {
            const mr_d2 x0 = X2[(int64_t)sc[p] * LPR], x1 = X2[(int64_t)sc[p + 1] * LPR], x2 = X2[(int64_t)sc[p + 2] * LPR],
                        x3 = X2[(int64_t)sc[p + 3] * LPR];
            const double v0 = sv[p], v1 = sv[p + 1], v2 = sv[p + 2], v3 = sv[p + 3];
            acc.x = fma(v0, x0.x, acc.x); acc.y = fma(v0, x0.y, acc.y);
            acc.x = fma(v1, x1.x, acc.x); acc.y = fma(v1, x1.y, acc.y);
            acc.x = fma(v2, x2.x, acc.x); acc.y = fma(v2, x2.y, acc.y);
            acc.x = fma(v3, x3.x, acc.x); acc.y = fma(v3, x3.y, acc.y);
        }
        for (; p < pe; ++p) {
            const mr_d2 x = X2[(int64_t)sc[p] * LPR];
            const double v = sv[p];
            acc.x = fma(v, x.x, acc.x);
            acc.y = fma(v, x.y, acc.y);
        }
    }
    if (!live) return;
    if (mk && mk[row] == 0.0) acc = reinterpret_cast<const mr_d2*>(alt)[row * LPR + jj];
    reinterpret_cast<mr_d2*>(Y)[row * LPR + jj] = acc;
}

// acc[t] = rows 16 t .. 16 t + 15 of  Ainv[nrow x n] R_sub[n x 16];  src = the slab [column][owned row] (owned dofs first in the list).
// 4 AM_CH columns (AM_CH matrix-core steps) at a time: their dof ids first, then the gathers of R in one flight, then the products.
// Measured at 117 649 subdomains of <= 138 x 24 (cfg 5's share), per sweep of sixteen columns: AM_CH 4: 0.64 ms (5 waves per SIMD),
// 8: 0.70 (4 waves), 16: 0.80 (3 waves): residency beats the longer flights
template <int RT, int AM_CH>
__device__ __forceinline__ void am_product(const double* __restrict__ src, const int32_t* __restrict__ ids, const double* __restrict__ R,
                                           int n, int nrow, int lk, int lj, mr_d4 (&acc)[RT]) {
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = mr_d4{0.0, 0.0, 0.0, 0.0};
    const int last = n * nrow - 1;
    for (int cb = 0; cb < n; cb += 4 * AM_CH) {     // (lists are NMAX long, entries past n hold dof 0: cb + 63 < NMAX)
        int32_t id[AM_CH];
        double bv[AM_CH];
#pragma unroll
        for (int s = 0; s < AM_CH; ++s) id[s] = ids[cb + 4 * s + lk];
#pragma unroll
        for (int s = 0; s < AM_CH; ++s) bv[s] = R[(int64_t)id[s] * MULTI_NR + lj];
#pragma unroll
        for (int s = 0; s < AM_CH; ++s) {
            const int c = cb + 4 * s + lk;
            if (cb + 4 * s < n) {       // (uniform over the wave)
                const bool on = c < n;
                const double b = on ? bv[s] : 0.0;
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    if (16 * t < nrow) {    // (uniform over the wave)
                        const int i = 16 * t + lj;
                        double a = src[min(c * nrow + i, last)];
                        a = (on && i < nrow) ? a : 0.0;
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
}

// A workgroup takes 4 AM_G consecutive places of the setup's order (subdomains with the same local matrix follow each other
// there), wave w the places base + w, base + 4 + w, ...; the slab of the first place's inverse is parked in LDS and serves
// every subdomain of the group that shares it (on a mesh with repeated cells: nearly all), the others read theirs from memory.
constexpr int AM_G = 4;
template <int RT, int AM_CH>   // owned rows of a subdomain <= 16 RT; AM_CH matrix-core steps (4 columns each) per flight of gathers
__global__ __launch_bounds__(256) void k_apply_multi(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_nown,
                                                     const int32_t* __restrict__ sub_dofs, const int64_t* __restrict__ inv_ptr,
                                                     const double* __restrict__ inv, const double* __restrict__ R,
                                                     double* __restrict__ Z, int32_t nsub, const int4* __restrict__ perm,
                                                     const double* __restrict__ mk, int lds_doubles) {
    constexpr int NR = MULTI_NR;
    extern __shared__ double sA[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lj = lane & 15, lk = lane >> 4;
    // XCD-contiguous ranges of places (neighbouring subdomains gather overlapping rows of R: one L2)
    const int nwg = gridDim.x, q_ = nwg >> 3, rem_ = nwg & 7, xcd_ = blockIdx.x & 7, within_ = blockIdx.x >> 3;
    const int wg = (xcd_ < rem_ ? xcd_ * (q_ + 1) : rem_ * (q_ + 1) + (xcd_ - rem_) * q_) + within_;
    const int32_t base = wg * (4 * AM_G);
    if (base >= nsub) return;
    const int32_t b0 = perm ? perm[base].x : base;
    const int64_t ptr0 = inv_ptr[b0];
    const int size0 = sub_n[b0] * sub_nown[b0];
    const bool parked = size0 <= lds_doubles;
    if (parked) {
        const double* __restrict__ s0 = inv + ptr0;     // (slabs start at multiples of 16 doubles and are padded to them)
        for (int f = 2 * (int)threadIdx.x; f < size0; f += 512) *reinterpret_cast<mr_d2*>(sA + f) = *reinterpret_cast<const mr_d2*>(s0 + f);
    }
    __syncthreads();
    for (int g = 0; g < AM_G; ++g) {
        const int32_t place = base + 4 * g + w;
        if (place >= nsub) return;
        const int32_t b = perm ? perm[place].x : place;
        const int n = sub_n[b], nrow = sub_nown[b];
        const int64_t ptr = inv_ptr[b];
        const bool from_lds = parked && ptr == ptr0;                // (uniform over the wave)
        const int32_t* __restrict__ ids = sub_dofs + (int64_t)b * NMAX;
        int32_t od[RT][4];      // the rows this lane stores: register q of tile t = owned row 16 t + lk + 4 q (column lj)
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) od[t][q] = ids[16 * t + lk + 4 * q];
        mr_d4 acc[RT];
        if (from_lds) am_product<RT, AM_CH>(sA, ids, R, n, nrow, lk, lj, acc);     // (two instances: LDS reads / global loads)
        else am_product<RT, AM_CH>(inv + ptr, ids, R, n, nrow, lk, lj, acc);
        // register q of tile t at lane (lk, lj): owned row 16 t + lk + 4 q, column lj
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * t + lk + 4 * q;
                if (i < nrow) {
                    const int32_t d = od[t][q];
                    double v = acc[t][q];
                    if (mk && mk[d] == 0.0) v = R[(int64_t)d * NR + lj];
                    Z[(int64_t)d * NR + lj] = v;
                }
            }
    }
}

}  // namespace

// can the stacked operators serve this context's preconditioner?  (restricted combination, the small-subdomain path)
bool multi_rhs_ok(const fedd_ctx* c) {
    return c->have_schwarz && !c->sw_big_active && c->sw_combine == FEDD_COMBINE_RESTRICTED && c->sw_max_own <= 128 && c->sw_nsub > 0 &&
           c->nnz > 0 && c->d_rowptr.p && c->d_val.p;
}

static int import_stacked(fedd_ctx* c, double* X) {
    if (c->n_cols == c->n_rows && c->halo.peers.empty()) return 0;
    HaloPlan& h = c->halo;
    FEDD_TRY(h.d_send_buf.ensure(h.send_lid.size() * (size_t)c->dofs * MULTI_NR));
    FEDD_TRY(h.d_recv_buf.ensure(h.recv_lid.size() * (size_t)c->dofs * MULTI_NR));
    return halo_import(c, X, c->dofs * MULTI_NR);
}

// Y = A X (owned rows), X with room for the ghost rows behind the owned ones (imported here); mk: Y = mk ? A X : alt
int spmm_owned(fedd_ctx* c, double* d_X, double* d_Y, const double* mk, const double* alt) {
    FEDD_TRY(import_stacked(c, d_X));
    constexpr int ROWS = 256 / (MULTI_NR / 2);
    hipLaunchKernelGGL(k_spmm, dim3((unsigned)((c->n_rows + ROWS - 1) / ROWS)), dim3(256), 0, c->stream, (const int32_t*)c->d_rowptr.p,
                       (const int32_t*)c->d_colind.p, (const double*)c->d_val.p, c->n_rows, (const double*)d_X, d_Y, mk, alt);
    FEDD_HIP(hipGetLastError());
    return 0;
}

// Z = M^-1 R (owned rows), R with room for the ghost rows (imported here); mk: Z = mk ? M^-1 R : R
int schwarz_apply_multi(fedd_ctx* c, double* d_R, double* d_Z, const double* mk) {
    FEDD_CHECK(multi_rhs_ok(c), "stacked Schwarz apply: needs the restricted one-level operator on subdomains of at most 128 owned dofs");
    FEDD_TRY(import_stacked(c, d_R));
    const int4* records = c->d_sw_order.p ? (const int4*)(c->d_sw_order.p + c->sw_order_off) : nullptr;
    const int32_t nsub = (int32_t)c->sw_nsub;
    const dim3 grid((unsigned)((nsub + 4 * AM_G - 1) / (4 * AM_G))), blk(256);
    // the shared slab in LDS while it fits 40 KB (three workgroups per CU)
    const int64_t slab = (((int64_t)c->sw_max_size * c->sw_max_own + 15) & ~(int64_t)15);
    const int lds_doubles = slab * 8 <= 40 * 1024 ? (int)slab : 0;
#define APPLY_MULTI1(RT, CH)                                                                                                       \
    hipLaunchKernelGGL((k_apply_multi<RT, CH>), grid, blk, (size_t)lds_doubles * sizeof(double), c->stream, (const int32_t*)c->d_sub_n.p, \
                       (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p, (const int64_t*)c->d_inv_ptr.p,       \
                       (const double*)c->d_inv.p, (const double*)d_R, d_Z, nsub, records, mk, lds_doubles)
#define APPLY_MULTI(RT)                           \
    if (c->multi_ch == 8) APPLY_MULTI1(RT, 8);    \
    else if (c->multi_ch == 16) APPLY_MULTI1(RT, 16); \
    else APPLY_MULTI1(RT, 4)
    if (c->sw_max_own <= 32) { APPLY_MULTI(2); }
    else if (c->sw_max_own <= 64) { APPLY_MULTI(4); }
    else if (c->sw_max_own <= 96) { APPLY_MULTI(6); }
    else { APPLY_MULTI(8); }
#undef APPLY_MULTI1
#undef APPLY_MULTI
    FEDD_HIP(hipGetLastError());
    return 0;
}

}  // namespace fedd
