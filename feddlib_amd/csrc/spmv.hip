// y = A x on the owned rows (CSR, f64 values, i32 local columns) with ghost import.
// Replaces Matrix::apply -> Xpetra/Tpetra CrsMatrix::apply, the SpMV Belos calls every GMRES
// iteration (feddlib/core/LinearAlgebra/Matrix_def.hpp:245-254; call chain
// feddlib/problems/Solver/LinearSolver_def.hpp:113-123).
//
// Kernel: row-per-lane-group ("CSR-vector"): LPR adjacent lanes share one row, so a wavefront
// streams 64/LPR consecutive rows whose (val, col) runs are contiguous in memory; x is gathered
// through L2 / Infinity Cache (8 B per lane), partial sums are combined with DPP shuffles.
// HBM-bound: algorithmic bytes = 12 nnz + 20 n_rows.
#include "fedd_internal.hpp"

namespace fedd {
namespace {

// optional epilogue of every SpMV kernel: y = A x - theta * sub (the shifted operator application of the s-step solver's Newton
// block basis, gmres.hip: one more vector read instead of a kernel of its own); sub == nullptr: y = A x
struct SpmvEpi {
    const double* sub;
    double theta;
};
__device__ __forceinline__ double epi_apply(const SpmvEpi& e, double s, int64_t r) { return e.sub ? s - e.theta * e.sub[r] : s; }

template <int LPR>
__global__ __launch_bounds__(256) void k_spmv(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                              const double* __restrict__ val, const double* __restrict__ x,
                                              double* __restrict__ y, int32_t n_rows, SpmvEpi epi) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t row = (int32_t)(gid / LPR);
    const int l = (int)(gid % LPR);
    double sum = 0.0;
    if (row < n_rows) {
        const int32_t e = rowptr[row + 1];
        for (int32_t p = rowptr[row] + l; p < e; p += LPR) sum += val[p] * x[colind[p]];
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) sum += __shfl_down(sum, off, LPR);
    if (l == 0 && row < n_rows) y[row] = epi_apply(epi, sum, row);
}

// "CSR-stream": a workgroup owns the rows whose first entry lies in its SP_CHUNK-wide window of
// the nonzero stream (window -> first row is a table built once per pattern by k_spmv_block_rows).
// All 256 lanes stream (val, col) of the window fully coalesced, gather x, park the products in
// LDS; then one lane per row adds its run.  No idle lanes for short rows, 8 independent loads per
// lane.  Workgroups are remapped so that each XCD (private L2) sweeps one contiguous eighth of
// the rows: its share of x then stays in that L2 instead of all of x passing through all eight.
constexpr int SP_CHUNK = 2048;

__global__ void k_spmv_block_rows(const int32_t* __restrict__ rowptr, int32_t n_rows, int32_t nb,
                                  int32_t* __restrict__ block_row, int chunk = SP_CHUNK) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    const int64_t target = (int64_t)b * chunk;
    int32_t lo = 0, hi = n_rows;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (rowptr[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    block_row[b] = lo;
}

// (non-temporal loads of the matrix stream were tried: slower in the solver, 41.9 vs 40.8 ms per step;
// the slabs of the Schwarz apply and the Krylov basis are the streams that are loaded non-temporally)
__global__ __launch_bounds__(256) void k_spmv_stream(const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ colind,
                                                     const double* __restrict__ val, const double* __restrict__ x,
                                                     double* __restrict__ y, const int32_t* __restrict__ block_row,
                                                     int32_t nb, SpmvEpi epi) {
    extern __shared__ double prod[];
    const int tid = threadIdx.x;
    // bijective XCD remap (blocks b and b+8 share an XCD): XCD k gets a contiguous range
    const int32_t q = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int32_t lb = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
    const int32_t R0 = block_row[lb], R1 = block_row[lb + 1];
    if (R0 >= R1) return;
    const int32_t base = rowptr[R0];
    const int32_t cnt = rowptr[R1] - base;
    // the row bounds of this lane's first row are requested before the value stream (memory returns
    // in issue order), so the row phase does not start with a dependent global load; clamped index:
    // the load is unconditional
    const int32_t r_first = min(R0 + tid, R1 - 1);
    const int32_t rb0 = rowptr[r_first], re0 = rowptr[r_first + 1];
    int32_t i = tid;
    for (; i + 7 * 256 < cnt; i += 8 * 256) {
        double v[8];
        int32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = val[base + i + u * 256];
            c[u] = colind[base + i + u * 256];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) prod[i + u * 256] = v[u] * x[c[u]];
    }
    for (; i < cnt; i += 256) prod[i] = val[base + i] * x[colind[base + i]];
    __syncthreads();
    for (int32_t r = R0 + tid; r < R1; r += 256) {
        const bool first = r == R0 + tid;
        const int32_t b = (first ? rb0 : rowptr[r]) - base, e = (first ? re0 : rowptr[r + 1]) - base;
        double s = 0.0;
        for (int32_t p = b; p < e; ++p) s += prod[p];
        y[r] = epi_apply(epi, s, r);
    }
}

// "CSR-window" (the default): like CSR-stream, but the window is the fixed slice [lb * CH, lb * CH + CH + ovh) of the
// nonzero stream, so the (val, col) loads depend on nothing but the kernel arguments: they are in
// flight while the window -> row table and the row bounds are still being fetched (CSR-stream needs
// block_row -> rowptr -> base before its first stream load: two dependent memory latencies per
// workgroup with nothing else in flight).  The `ovh` entries past the window hold the tail of the
// last row that starts inside it (ovh >= max_row_nnz); the head of the window up to rowptr[R0]
// belongs to the previous workgroup's last row and is loaded but not used.
// Measured (50 launches back to back): 100^3 cells 36.4 -> 31.0 us, 214^3 cells 437 -> 385 us.  Tried on top
// and dropped: 16-byte loads (same), windows of 6 / 10 / 12 x 256 entries (same within 2 %), a persistent
// variant that prefetches the next window behind the gathers (87 VGPRs, 5 workgroups per CU: slower),
// gathering in the row phase with 1 / 2 / 4 lanes per row (coalesced gathers: same or slower).
// Non-temporal loads of the matrix stream (option "spmv_nt"): 214^3 cells 406 -> 380 us back to back (x is no
// longer pushed out of L2 between the visits of neighbouring node planes) but within the noise inside the solver
// (same box: 391 / 376 us and 395 / 394 us, step 636 / 634 and 638 / 638 ms); 100^3 cells 31 -> 40 us (the 203 MB
// matrix is partly served by the Infinity Cache from one SpMV to the next): by default on for matrices larger than
// that cache (spmv_nt = -1).  Also tried on the window kernel: gathering in the row phase from LDS-staged (val, col)
// with 1 / 2 lanes per row (376 -> 386 / 391 us at 214^3): the gathers are not what limits it; in real traffic
// (2.14 GB per launch, x fetched three times) the kernel runs at 0.91 of the read ceiling.
// NU = entries per lane: the window is 256 NU entries (8 = SP_CHUNK by default; option "spmv_win_nu" for the compacted stream).
// C16: the columns as 16-bit offsets from the smallest column of the entry's own window (col16, wbase; k_cs_col16): 10 instead
// of 12 bytes per entry in every window whose columns span less than 65536 (every mesh numbered with some locality; the windows
// that reach ghost columns on several ranks, or far neighbours, keep 32-bit indices: wbase < 0)
template <bool NT, int NU = SP_CHUNK / 256, bool C16 = false>
__global__ __launch_bounds__(256) void k_spmv_win(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                  const double* __restrict__ val, const double* __restrict__ x,
                                                  double* __restrict__ y, const int32_t* __restrict__ block_row,
                                                  int32_t nb, int32_t nnz, int32_t ovh, SpmvEpi epi,
                                                  const uint16_t* __restrict__ col16 = nullptr,
                                                  const int32_t* __restrict__ wbase = nullptr) {
    extern __shared__ double prod[];
    constexpr int CH = 256 * NU;
    const int tid = threadIdx.x;
    const int32_t q = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int32_t lb = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
    const int32_t R0 = block_row[lb], R1 = block_row[lb + 1];
    const int32_t base = lb * CH, last = nnz - 1;
    int32_t cb0 = 0, cb1 = 0;       // column bases of this window and of the next one (the overhang's entries belong to it)
    if (C16) {
        cb0 = wbase[lb];
        cb1 = wbase[min(lb + 1, nb - 1)];
    }
    double v[NU + 1];
    int32_t cc[NU + 1];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int32_t idx = min(base + tid + u * 256, last);
        v[u] = NT ? __builtin_nontemporal_load(val + idx) : val[idx];
        // the 16-bit offset is requested unconditionally, with the value: its address depends on nothing.  (cb0 < 0: this window's
        // columns span too much for 16 bits and its entries take their 32-bit indices -- a second, dependent load, on the rare
        // path only; loading the offsets behind the test of cb0 put the base's latency in front of every gather: 26.8 -> 34 us
        // at the share of cfg 3)
        if (C16) {
            const int32_t c16v = (int32_t)(NT ? __builtin_nontemporal_load(col16 + idx) : col16[idx]);
            cc[u] = cb0 + c16v;
        } else {
            cc[u] = NT ? __builtin_nontemporal_load(colind + idx) : colind[idx];
        }
    }
    if (C16 && cb0 < 0) {   // (uniform over the workgroup)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int32_t idx = min(base + tid + u * 256, last);
            cc[u] = NT ? __builtin_nontemporal_load(colind + idx) : colind[idx];
        }
    }
    v[NU] = 0.0;
    cc[NU] = 0;
    if (tid < ovh) {   // the overhang: tail of the last row that starts inside the window
        const int32_t idx = min(base + CH + tid, last);
        v[NU] = NT ? __builtin_nontemporal_load(val + idx) : val[idx];
        if (C16) {
            const int32_t cbo = idx >= base + CH ? cb1 : cb0;   // (the clamp at the end of the stream stays inside this window)
            const int32_t c16v = (int32_t)(NT ? __builtin_nontemporal_load(col16 + idx) : col16[idx]);
            cc[NU] = cbo + c16v;
            if (cbo < 0) cc[NU] = NT ? __builtin_nontemporal_load(colind + idx) : colind[idx];
        } else {
            cc[NU] = NT ? __builtin_nontemporal_load(colind + idx) : colind[idx];
        }
    }
    // row bounds of the lane's first TWO rows, requested ahead of the gathers: with the compacted stream (7 entries
    // per interior row) a window holds ~290 rows, so the second trip of the row phase is the normal case and
    // would otherwise start with a dependent global load (186 -> 182 us on the 214^3 grid)
    const int32_t r_a = max(min(R0 + tid, R1 - 1), 0), r_b = max(min(R0 + tid + 256, R1 - 1), 0);
    const int32_t rb0 = rowptr[r_a], re0 = rowptr[r_a + 1];
    const int32_t rb1 = rowptr[r_b], re1 = rowptr[r_b + 1];
    double xg[NU + 1];
#pragma unroll
    for (int u = 0; u <= NU; ++u) xg[u] = x[cc[u]];
#pragma unroll
    for (int u = 0; u < NU; ++u) prod[tid + u * 256] = v[u] * xg[u];
    if (tid < ovh) prod[CH + tid] = v[NU] * xg[NU];
    __syncthreads();
    int trip = 0;
    for (int32_t r = R0 + tid; r < R1; r += 256, ++trip) {
        const int32_t b = (trip == 0 ? rb0 : trip == 1 ? rb1 : rowptr[r]) - base;
        const int32_t e = (trip == 0 ? re0 : trip == 1 ? re1 : rowptr[r + 1]) - base;
        double s = 0.0;
        for (int32_t p = b; p < e; ++p) s += prod[p];
        y[r] = epi_apply(epi, s, r);
    }
}

// ---- compacted stream: the owned rows without their (numerically) zero entries ----------------------------------
// Dropped: entries with |a_ij| <= tol * max_k |a_ik| (per row).  tol = 0 drops exactly the entries that are 0.0
// (then y is bit for bit the y of the parity CSR for finite x); the default tol = 2^-52 also drops what is below one
// ulp of the row's largest entry: on the Kuhn-split cube 8 of the 15 pattern entries of an interior Laplace row are
// structural zeros, of which floating-point cancellation leaves 4 as exact 0.0 and 4 as +-1e-17 * diag noise (the
// reference's own assembly has such noise too, and an optional threshold for it: setZeros_, FE_def.hpp:719-721).
// Each dropped product is below the rounding error of the row sum itself.
// Same windows as the SpMV: workgroup b owns the rows whose first entry lies in window b of the parity CSR, i.e. the
// contiguous entry range [rowptr[R0], rowptr[R1]).  Pass 1 counts the kept entries, a scan over the windows gives each
// its offset in the compacted stream, pass 2 compacts the range through an LDS prefix sum (coalesced loads and stores)
// and writes the new row starts.  ~3.8 GB of traffic for the 214^3 grid, once per assembled matrix.
constexpr int CS_OVH = 256;   // windowed SpMV is only used for rows of at most 256 entries

// values of the window's entry range -> sval, keep flags -> sflag (one lane per row decides its entries)
__device__ __forceinline__ void cs_window_flags(const int32_t* __restrict__ rowptr, const double* __restrict__ val,
                                                int32_t R0, int32_t R1, int32_t lo, int32_t len, double tol,
                                                double* sval, uint8_t* sflag, int tid) {
    {
        // (all loads of the window in flight before the first store to LDS: a load -> store loop waits for memory once per trip)
        constexpr int U = (SP_CHUNK + CS_OVH + 255) / 256;
        double rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int32_t j = tid + 256 * u;
            rv[u] = j < len ? __builtin_nontemporal_load(val + lo + j) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int32_t j = tid + 256 * u;
            if (j < len) sval[j] = rv[u];
        }
    }
    __syncthreads();
    for (int32_t r = R0 + tid; r < R1; r += 256) {
        const int32_t b = rowptr[r] - lo, e = rowptr[r + 1] - lo;
        double mx = 0.0;
        for (int32_t p = b; p < e; ++p) mx = fmax(mx, fabs(sval[p]));
        const double thr = tol * mx;
        for (int32_t p = b; p < e; ++p) sflag[p] = !(fabs(sval[p]) <= thr) ? 1 : 0;   // a NaN stays in the stream
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_cs_count(const int32_t* __restrict__ rowptr, const double* __restrict__ val,
                                                  const int32_t* __restrict__ block_row, int32_t nb, double tol,
                                                  int32_t* __restrict__ wincnt) {
    __shared__ double sval[SP_CHUNK + CS_OVH];
    __shared__ uint8_t sflag[SP_CHUNK + CS_OVH];
    __shared__ int32_t red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t R0 = block_row[b], R1 = block_row[b + 1];
    int32_t cnt = 0;
    if (R0 < R1) {   // (uniform over the workgroup)
        const int32_t lo = rowptr[R0], len = rowptr[R1] - lo;
        cs_window_flags(rowptr, val, R0, R1, lo, len, tol, sval, sflag, tid);
        for (int32_t j = tid; j < len; j += 256) cnt += sflag[j];
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) wincnt[b] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_cs_fill(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                 const double* __restrict__ val, const int32_t* __restrict__ block_row,
                                                 const int32_t* __restrict__ winoff, double tol,
                                                 int32_t* __restrict__ cs_rowptr,
                                                 int32_t* __restrict__ cs_col, double* __restrict__ cs_val) {
    __shared__ double sval[SP_CHUNK + CS_OVH];
    __shared__ uint8_t sflag[SP_CHUNK + CS_OVH];
    __shared__ int32_t pre[SP_CHUNK + CS_OVH + 1];
    __shared__ int32_t wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t R0 = block_row[b], R1 = block_row[b + 1];
    if (R0 >= R1) return;
    const int32_t lo = rowptr[R0], hi = rowptr[R1], len = hi - lo;   // len <= SP_CHUNK + max_row_nnz - 1
    const int32_t wbase = winoff[b];
    cs_window_flags(rowptr, val, R0, R1, lo, len, tol, sval, sflag, tid);
    // exclusive prefix of the keep flags: lane t takes the K consecutive positions [t K, t K + K)
    const int K = (len + 255) / 256;
    int32_t mine = 0;
    for (int k = 0; k < K; ++k) {
        const int32_t j = tid * K + k;
        mine += j < len ? sflag[j] : 0;
    }
    int32_t inc = mine;   // inclusive scan inside the wave
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t up = __shfl_up(inc, off, 64);
        if ((tid & 63) >= off) inc += up;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    int32_t run = inc - mine;
    for (int w = 0; w < (tid >> 6); ++w) run += wsum[w];
    for (int k = 0; k < K; ++k) {
        const int32_t j = tid * K + k;
        if (j < len) {
            pre[j] = run;
            run += sflag[j];
        }
    }
    if (tid == 255) pre[len] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    {
        constexpr int U = (SP_CHUNK + CS_OVH + 255) / 256;
        int32_t rc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int32_t j = tid + 256 * u;
            rc[u] = (j < len && sflag[j]) ? __builtin_nontemporal_load(colind + lo + j) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int32_t j = tid + 256 * u;
            if (j < len && sflag[j]) {
                cs_val[wbase + pre[j]] = sval[j];
                cs_col[wbase + pre[j]] = rc[u];
            }
        }
    }
    for (int32_t r = R0 + tid; r < R1; r += 256) cs_rowptr[r] = wbase + pre[rowptr[r] - lo];
}

// 16-bit columns of the compacted stream for k_spmv_win<.., C16>: window lb (CH entries of the stream) takes the smallest column
// among its entries as base, or -1 if it spans 65536 columns or more (its entries are then read from the 32-bit array; flag counts them)
__global__ __launch_bounds__(256) void k_cs_col16(const int32_t* __restrict__ col, int32_t total, int32_t CH,
                                                  int32_t* __restrict__ wbase, uint16_t* __restrict__ col16, int32_t* __restrict__ flag) {
    __shared__ int32_t smin[4], smax[4];
    const int tid = threadIdx.x;
    const int32_t lo = blockIdx.x * CH, hi = min(total, lo + CH);
    int32_t mn = 0x7fffffff, mx = 0;
    for (int32_t i = lo + tid; i < hi; i += 256) {
        const int32_t cidx = col[i];
        mn = min(mn, cidx);
        mx = max(mx, cidx);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = min(mn, __shfl_xor(mn, off, 64));
        mx = max(mx, __shfl_xor(mx, off, 64));
    }
    if ((tid & 63) == 0) {
        smin[tid >> 6] = mn;
        smax[tid >> 6] = mx;
    }
    __syncthreads();
    mn = min(min(smin[0], smin[1]), min(smin[2], smin[3]));
    mx = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
    const bool wide = lo < hi && mx - mn > 65535;
    if (lo >= hi) mn = 0;
    if (tid == 0) {
        wbase[blockIdx.x] = wide ? -1 : mn;
        if (wide) atomicAdd(flag, hi - lo);     // entries that keep 32-bit columns
    }
    if (!wide)
        for (int32_t i = lo + tid; i < hi; i += 256) col16[i] = (uint16_t)(col[i] - mn);
}

// Read-only streaming calibration (fedd_read_bandwidth): sums `n2` double2 with 16-byte loads, four independent
// loads per lane in flight, grid-stride.  What the box's HBM delivers to a pure read stream; the byte models of
// SpMV / Schwarz apply are quoted against the 8 TB/s spec AND against this.
typedef double vd2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_read_stream(const vd2* __restrict__ p, size_t n2, double* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    double s = 0.0;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        const vd2 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride),
                  c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        s += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    for (; i < n2; i += stride) s += p[i].x + p[i].y;
    if (s == 12345.678) out[0] = s;
}

}  // namespace

int read_bandwidth(fedd_ctx* c, int64_t bytes, int reps, double* gbs) {
    DevBuf<double> buf, out;
    const size_t n = (size_t)bytes / 8;
    FEDD_TRY(buf.ensure(n));
    FEDD_TRY(out.ensure(1));
    FEDD_HIP(hipMemsetAsync(buf.p, 0, n * 8, c->stream));
    hipEvent_t a, b;
    FEDD_HIP(hipEventCreate(&a));
    FEDD_HIP(hipEventCreate(&b));
    float best = 1e30f;
    for (int pass = 0; pass < 3; ++pass) {   // best of three batches (the first one also warms up)
        FEDD_HIP(hipEventRecord(a, c->stream));
        for (int k = 0; k < reps; ++k)
            hipLaunchKernelGGL(k_read_stream, dim3(8192), dim3(256), 0, c->stream, (const vd2*)buf.p, n / 2, out.p);
        FEDD_HIP(hipEventRecord(b, c->stream));
        FEDD_HIP(hipEventSynchronize(b));
        float ms = 0.f;
        FEDD_HIP(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *gbs = (double)(n * 8) * reps / ((double)best * 1e6);
    return 0;
}

// ---- column patterns of the compacted stream ---------------------------------------------------------------------
// On a mesh with repeated cells the rows of the matrix repeat their column OFFSETS (col - row): the 214^3 cube has a
// few dozen distinct offset lists among its 9.9 M rows.  The solver's stream then needs no column index per entry: a
// 16-bit pattern id per row and a small table of offset lists (in LDS) give the columns, the values stay per row, exact.
// 12 -> 8 bytes per entry.  Rows whose list is not in the table (more than SPAT_L entries, or the table full) keep
// their explicit columns (id SPAT_EXPL); more than SPAT_P patterns (an unstructured mesh): the dictionary is not used.
// The sum of a row runs over its entries in order with separate multiply and add, exactly like k_spmv_win: same bits.
constexpr int SPAT_P = 64, SPAT_L = 48, SPAT_TS = 1024;
constexpr int SPAT_LK = 16;     // longest pattern k_spmv_pat unrolls; longer patterns (vector problems: 3 x 15 entries per row) exist for the
                                // row classes' sake only (k_spmv_cls) -- without classes such a matrix goes back to the per-entry kernel
constexpr uint16_t SPAT_EXPL = 0xffff;

__device__ __forceinline__ uint64_t sp_mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// 8 lanes per row: hash of (length, offsets in order); 0 = the row keeps explicit columns
__global__ void k_pat_hash(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int32_t n,
                           unsigned long long* __restrict__ hash) {
    const int32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int e = threadIdx.x & 7;
    unsigned long long h = 0;
    int len = 0;
    if (r < n) {
        const int32_t b = rowptr[r];
        len = rowptr[r + 1] - b;
        if (len <= SPAT_L)
            for (int j = e; j < len; j += 8)
                h += sp_mix(((unsigned long long)(j + 1) << 40) ^ (unsigned long long)(uint32_t)(col[b + j] - r));
    }
    for (int off = 4; off > 0; off >>= 1) h += __shfl_xor(h, off, 8);
    if (r < n && e == 0) {
        h = sp_mix(h + (unsigned long long)len);
        hash[r] = (len >= 1 && len <= SPAT_L) ? (h | 1ull) : 0ull;
    }
}

// table insert (the lanes of a wave that carry the same hash send one of them); the slot keeps the lowest row seen
__global__ void k_pat_insert(const unsigned long long* __restrict__ hash, int32_t n, unsigned long long* __restrict__ tkey,
                             int32_t* __restrict__ tmin, int32_t* __restrict__ slot_of, int32_t* __restrict__ n_claimed) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long h = r < n ? hash[r] : 0ull;
    const bool active = h != 0ull;
    uint64_t todo = __ballot(active);
    int32_t mine = -1;
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const unsigned long long lh = __shfl(h, leader, 64);
        const uint64_t same = __ballot(active && h == lh) & todo;
        int32_t s_found = -1;
        // (many more distinct lists than the table is meant for -- an unstructured mesh --: stop looking, the rows keep
        // their explicit columns and the dictionary ends up unused)
        if (lane == leader && __hip_atomic_load(n_claimed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 4 * SPAT_P) {
            int s = (int)(lh % SPAT_TS);
            for (int probe = 0; probe < 64; ++probe) {
                // plain look first: after the first waves the key is there and its lowest row is below this one, and
                // hundreds of thousands of atomics on a handful of addresses would serialise in L2
                unsigned long long old = __hip_atomic_load(&tkey[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == 0ull) {
                    old = atomicCAS(&tkey[s], 0ull, lh);
                    if (old == 0ull) atomicAdd(n_claimed, 1);
                }
                if (old == 0ull || old == lh) {
                    if (__hip_atomic_load(&tmin[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > r) atomicMin(&tmin[s], r);
                    s_found = s;
                    break;
                }
                s = s + 1 == SPAT_TS ? 0 : s + 1;
            }
        }
        s_found = __shfl(s_found, leader, 64);
        if ((same >> lane) & 1ull) mine = s_found;
        todo &= ~same;
    }
    if (r < n) slot_of[r] = mine;
}

// one workgroup: pattern ids in slot order, offset lists from the representative rows
__global__ __launch_bounds__(256) void k_pat_table(const unsigned long long* __restrict__ tkey, const int32_t* __restrict__ tmin,
                                                   const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   int32_t* __restrict__ pat_of_slot, int32_t* __restrict__ plen,
                                                   int32_t* __restrict__ pdelta, int32_t* __restrict__ n_pat) {
    __shared__ int32_t cnt[SPAT_TS];
    const int tid = threadIdx.x;
    for (int s = tid; s < SPAT_TS; s += 256) cnt[s] = tkey[s] != 0ull ? 1 : 0;
    __syncthreads();
    if (tid == 0) {         // (a thousand slots: a serial scan is microseconds)
        int run = 0;
        for (int s = 0; s < SPAT_TS; ++s) {
            const int c = cnt[s];
            cnt[s] = c ? run : -1;
            run += c;
        }
        *n_pat = run;
        n_pat[3] = 0;       // (longest pattern, below)
    }
    __syncthreads();
    for (int s = tid; s < SPAT_TS; s += 256) {
        const int id = cnt[s];
        pat_of_slot[s] = id < SPAT_P ? id : -1;
        if (id >= 0 && id < SPAT_P) {
            const int32_t r = tmin[s], b = rowptr[r], len = rowptr[r + 1] - b;
            plen[id] = len;
            atomicMax(n_pat + 3, len);
            for (int j = 0; j < SPAT_L; ++j) pdelta[id * SPAT_L + j] = j < len ? col[b + j] - r : 0;
        }
    }
}

// per row: its pattern id, checked entry by entry against the table (a colliding hash costs explicit columns, never a
// wrong product); n_expl counts the rows that keep explicit columns
__global__ void k_pat_rows(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int32_t n,
                           const int32_t* __restrict__ slot_of, const int32_t* __restrict__ pat_of_slot,
                           const int32_t* __restrict__ plen, const int32_t* __restrict__ pdelta, uint16_t* __restrict__ pat,
                           int32_t* __restrict__ n_expl, const uint16_t* prev, int32_t n_prev_pat) {
    // 8 lanes per row (consecutive lanes read consecutive column ids)
    // prev != nullptr (may be pat itself): the candidate of a row is its id of the PREVIOUS matrix' dictionary (slot_of /
    // pat_of_slot are not read): the dictionary is kept while every row that had a pattern still matches it -- n_expl[3] counts
    // the rows that do not (any: the caller rebuilds)
    const int32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int e = threadIdx.x & 7;
    int id = -1, len = 0;
    bool ok = true;
    bool had = false;
    if (r < n) {
        if (prev) {
            const uint16_t pv = prev[r];
            id = (pv != SPAT_EXPL && (int)pv < n_prev_pat) ? (int)pv : -1;
            had = pv != SPAT_EXPL;
        } else {
            const int32_t s = slot_of[r];
            id = s >= 0 ? pat_of_slot[s] : -1;
        }
        const int32_t b = rowptr[r];
        len = rowptr[r + 1] - b;
        if (id >= 0) {
            ok = len == plen[id];
            if (ok)
                for (int j = e; j < len; j += 8) ok = ok && (col[b + j] - r == pdelta[id * SPAT_L + j]);
        }
    }
    unsigned bad = ok ? 0u : 1u;
    for (int off = 4; off > 0; off >>= 1) bad |= __shfl_xor(bad, off, 8);
    if (bad) id = -1;
    const bool writer = r < n && e == 0;
    if (writer) pat[r] = id >= 0 ? (uint16_t)id : SPAT_EXPL;
    if (prev) {
        const uint64_t lost = __ballot(writer && had && id < 0);
        if (lost && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(lost)) atomicAdd(n_expl + 3, (int32_t)__builtin_popcountll(lost));
    }
    // one atomic per wave for the count (every row of an unstructured mesh lands here)
    const uint64_t expl = __ballot(writer && id < 0);
    if (expl && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(expl)) atomicAdd(n_expl, (int32_t)__builtin_popcountll(expl));
    int wmax = writer ? len : 0;                                    // longest compacted row: one atomic per wave, if at all
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off, 64));
    if ((threadIdx.x & 63) == 0 && wmax > __hip_atomic_load(n_expl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(n_expl + 1, wmax);
}

// The windowed SpMV on the value stream alone: workgroup lb brings the values [CH lb, CH lb + CH + overhang) to LDS
// (coalesced, non-temporal, addresses independent of any other load), then one lane per row multiplies its values with x
// at row + offset -- consecutive lanes are consecutive rows of (mostly) one pattern, so the reads of x are coalesced,
// which the per-entry gathers of k_spmv_win are not.  CH = 256 NU with NU = the usual row length: a window then holds
// about 256 rows, one per lane.  (A variant with 256 ROWS per workgroup was slower, 174 against 152 us at 214^3: its value
// loads depend on two row pointers.)
template <bool NT, int NU, int LU /* unrolled entries per row: 8 when no pattern is longer, else SPAT_LK */>
__global__ __launch_bounds__(256) void k_spmv_pat(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                  const double* __restrict__ val, const uint16_t* __restrict__ pat,
                                                  const int32_t* __restrict__ plen, const int32_t* __restrict__ pdelta,
                                                  int32_t n_pat, const double* __restrict__ x, double* __restrict__ y,
                                                  const int32_t* __restrict__ block_row, int32_t nb, int32_t nnz, int32_t ovh,
                                                  SpmvEpi epi) {
    extern __shared__ double sval[];                    // [CH + ovh]
    __shared__ int32_t sdelta[SPAT_P * SPAT_LK];      // (this kernel runs patterns of at most SPAT_LK entries)
    __shared__ int32_t slen[SPAT_P];
    constexpr int CH = 512 * NU;                        // NU 16-byte loads per lane (the value array is padded by a window)
    const int tid = threadIdx.x;
    const int32_t q = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int32_t lb = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
    const int32_t R0 = block_row[lb], R1 = block_row[lb + 1];
    const int32_t base = lb * CH, last = nnz - 1;
    vd2 v[NU];
    const vd2* __restrict__ val2 = reinterpret_cast<const vd2*>(val + base);
#pragma unroll
    for (int u = 0; u < NU; ++u) v[u] = NT ? __builtin_nontemporal_load(val2 + tid + u * 256) : val2[tid + u * 256];
    double vo = 0.0;
    if (tid < ovh) {
        const int32_t idx = min(base + CH + tid, last);
        vo = NT ? __builtin_nontemporal_load(val + idx) : val[idx];
    }
    // the lane's first two rows: bounds and pattern ids requested with the stream
    const int32_t r_a = max(min(R0 + tid, R1 - 1), 0), r_b = max(min(R0 + tid + 256, R1 - 1), 0);
    const int32_t rb0 = rowptr[r_a], rb1 = rowptr[r_b];
    const uint16_t id0 = pat[r_a], id1 = pat[r_b];
    for (int i = tid; i < n_pat * SPAT_LK; i += 256) sdelta[i] = pdelta[(i / SPAT_LK) * SPAT_L + (i % SPAT_LK)];
    for (int i = tid; i < n_pat; i += 256) slen[i] = plen[i];
#pragma unroll
    for (int u = 0; u < NU; ++u) reinterpret_cast<vd2*>(sval)[tid + u * 256] = v[u];
    if (tid < ovh) sval[CH + tid] = vo;
    __syncthreads();
    int trip = 0;
    for (int32_t r = R0 + tid; r < R1; r += 256, ++trip) {
        const int32_t rb = trip == 0 ? rb0 : trip == 1 ? rb1 : rowptr[r];
        const int32_t b = rb - base;
        const uint16_t id = trip == 0 ? id0 : trip == 1 ? id1 : pat[r];
        double s = 0.0;
        {
            // separate multiply and add, in entry order: the bits of k_spmv_win (HIP contracts a * b + c by default, also
            // through __dmul_rn / __dadd_rn, which are plain operators)
#pragma clang fp contract(off)
            if (id != SPAT_EXPL) {
                const int len = slen[id];
                const int32_t* __restrict__ dl = sdelta + (int)id * SPAT_LK;
                double xv[LU], av[LU];
#pragma unroll
                for (int j = 0; j < LU; ++j) {          // all loads of the row in flight together
                    const bool on = j < len;
                    xv[j] = on ? x[r + dl[j]] : 0.0;
                    av[j] = on ? sval[b + j] : 0.0;
                }
#pragma unroll
                for (int j = 0; j < LU; ++j)
                    if (j < len) {
                        const double pr = av[j] * xv[j];
                        s = s + pr;
                    }
            } else {
                const int32_t e = rowptr[r + 1] - base;
                for (int32_t p = b; p < e; ++p) {
                    const double pr = sval[p] * x[colind[base + p]];
                    s = s + pr;
                }
            }
        }
        y[r] = epi_apply(epi, s, r);
    }
}

// ------------------------------------------------------------------------------------------------
// Row classes (round 4).  On a structured grid not only the column offsets of the rows repeat, their VALUES do too: the
// assembled Laplace matrix of the 214^3-cell cube has 2 174 bitwise-distinct rows among its 9.66 million interior ones
// (tools/experiments/row_classes.py; they differ in the last bits, by where the coordinates i h round).  Rows with the same
// column pattern AND the same values, bit for bit, form a class; the solver's SpMV then reads a 4-byte (class, pattern) word per row and
// takes the row's values from a table of at most CLS_MAX classes (64 bytes each: L2 resident, the frequent ones in L1) instead of streaming 8
// bytes per entry -- the same products in the same order, so y is the same bit for bit (a row whose hash collides keeps its
// stream entries: every classed row is verified entry by entry against the table).  Built with the column patterns, per
// assembled matrix: k_cls_insert -> k_cls_table -> k_cls_rows.
// ------------------------------------------------------------------------------------------------
constexpr int CLS_L = SPAT_L;       // longest row that can join a class; the table stride is the longest pattern rounded up to 8
constexpr int CLS_MAX = 16384;      // classes (16-bit ids; their pattern ids live in LDS: 32 KB)
constexpr int CLS_TS = 65536;       // hash table slots
constexpr uint32_t CLS_NONE = 0xffffffffu;     // per row: class << 8 | column pattern, or none

// hash of a row's (pattern, length, value bits) and its slot in the table.  Eight lanes per row read its values (consecutive
// lanes, consecutive entries: coalesced; a lane per row reading 56 bytes at a 56-byte stride ran at 1 TB/s) and add their
// position-keyed mixes (order independent, like k_pat_hash); lane 0 of the row probes, on its own: ordinary cached loads first
// -- a key never changes once it is set, so a non-zero value seen is right and a stale zero is resolved by the compare-and-swap
// (atomic loads of the dozen hot keys would all queue at their L2 channels) --, a compare-and-swap only for an empty slot; the
// slot keeps the lowest row seen (its representative).  (A leader-per-distinct-hash loop as in k_pat_insert serialises here: a
// wave of 64 consecutive rows holds a dozen classes, not one or two patterns.)
__global__ __launch_bounds__(256) void k_cls_insert(const int32_t* __restrict__ rowptr, const double* __restrict__ val,
                                                    const uint16_t* __restrict__ pat, int32_t n, unsigned long long* __restrict__ tkey,
                                                    int32_t* __restrict__ tmin, int32_t* __restrict__ slot_of,
                                                    int32_t* __restrict__ n_claimed) {
    const int32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int e = threadIdx.x & 7;
    unsigned long long h = 0ull;
    int32_t len = 0;
    uint16_t id = SPAT_EXPL;
    if (r < n) {
        const int32_t b = rowptr[r];
        len = rowptr[r + 1] - b;
        id = pat[r];
        if (id != SPAT_EXPL && len >= 1 && len <= CLS_L)
            for (int j = e; j < len; j += 8)
                h += sp_mix(((unsigned long long)(j + 1) * 0x9E3779B97F4A7C15ull) ^ (unsigned long long)__double_as_longlong(val[b + j]));
    }
    for (int off = 4; off > 0; off >>= 1) h += __shfl_xor(h, off, 8);
    if (r >= n || e != 0) return;
    int32_t found = -1;
    if (id != SPAT_EXPL && len >= 1 && len <= CLS_L) {
        h = sp_mix(h + (((unsigned long long)id << 8) | (unsigned long long)len)) | 1ull;
        if (*n_claimed <= 2 * CLS_MAX) {
            int s2 = (int)(h % CLS_TS);
            for (int probe = 0; probe < 128; ++probe) {
                unsigned long long old = tkey[s2];
                if (old == 0ull) {
                    old = atomicCAS(&tkey[s2], 0ull, h);
                    if (old == 0ull) atomicAdd(n_claimed, 1);
                }
                if (old == 0ull || old == h) {
                    if (tmin[s2] > r) atomicMin(&tmin[s2], r);
                    found = s2;
                    break;
                }
                s2 = s2 + 1 == CLS_TS ? 0 : s2 + 1;
            }
        }
    }
    slot_of[r] = found;
}

// one workgroup: class ids in slot order
__global__ __launch_bounds__(1024) void k_cls_table(const unsigned long long* __restrict__ tkey, int32_t* __restrict__ cls_of_slot,
                                                    int32_t* __restrict__ n_cls) {
    __shared__ int32_t part[1024];
    const int tid = threadIdx.x;
    constexpr int PER = CLS_TS / 1024;
    int cnt = 0;
    for (int k = 0; k < PER; ++k) cnt += tkey[tid * PER + k] != 0ull ? 1 : 0;
    part[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int t = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    int run = part[tid] - cnt;
    if (tid == 1023) *n_cls = part[1023];
    for (int k = 0; k < PER; ++k) {
        const int sl = tid * PER + k;
        int id = -1;
        if (tkey[sl] != 0ull) {
            id = run < CLS_MAX ? run : -1;
            ++run;
        }
        cls_of_slot[sl] = id;
    }
}

// a thread per slot: values and pattern id of a class from its representative row (the lowest row of the slot)
__global__ void k_cls_fill(const int32_t* __restrict__ cls_of_slot, const int32_t* __restrict__ tmin, const int32_t* __restrict__ rowptr,
                           const double* __restrict__ val, const uint16_t* __restrict__ pat, double* __restrict__ cls_val,
                           uint16_t* __restrict__ cls_pat, int cl) {
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= CLS_TS) return;
    const int id = cls_of_slot[sl];
    if (id < 0) return;
    const int32_t r = tmin[sl], b = rowptr[r], len = rowptr[r + 1] - b;
    for (int j = 0; j < cl; ++j) cls_val[(size_t)id * cl + j] = j < len ? val[b + j] : 0.0;
    cls_pat[id] = pat[r];
}

// per row (eight lanes): its class, verified bit for bit against the table (differences OR-ed, no short-circuit: all loads fly
// together); counters: [0] classed rows, [1] stream entries of the others
__global__ __launch_bounds__(256) void k_cls_rows(const int32_t* __restrict__ rowptr, const double* __restrict__ val,
                                                  const uint16_t* __restrict__ pat, int32_t n, const int32_t* __restrict__ slot_of,
                                                  const int32_t* __restrict__ cls_of_slot, const double* __restrict__ cls_val,
                                                  const uint16_t* __restrict__ cls_pat, uint32_t* __restrict__ cls,
                                                  int32_t* __restrict__ counters, int cl) {
    // (grid-stride over the rows, the two counts summed per lane, per wave, per workgroup: ONE atomic pair per workgroup -- an
    // atomic per wave on one address was 14 of the kernel's 14.1 ms at 214^3 cells)
    __shared__ int32_t s_in[4], s_rest[4];
    const int e = threadIdx.x & 7;
    int my_in = 0, my_rest = 0;
    for (int64_t r64 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; r64 < n; r64 += ((int64_t)gridDim.x * blockDim.x) >> 3) {
        const int32_t r = (int32_t)r64;
        const int32_t b = rowptr[r], len = rowptr[r + 1] - b;
        const int32_t sl = slot_of[r];
        const int32_t id = sl >= 0 ? cls_of_slot[sl] : -1;
        const uint16_t pid = pat[r];
        const bool cand = id >= 0 && len <= cl && cls_pat[id] == pid;
        long long diff = 0;
        if (cand)
            for (int j = e; j < len; j += 8) diff |= __double_as_longlong(val[b + j]) ^ __double_as_longlong(cls_val[(size_t)id * cl + j]);
        for (int off = 4; off > 0; off >>= 1) diff |= __shfl_xor(diff, off, 8);
        const bool in = cand && diff == 0;      // (the padding of a table entry is zero by construction: k_cls_fill)
        if (e == 0) {
            cls[r] = in ? ((uint32_t)id << 8) | (uint32_t)pid : CLS_NONE;
            my_in += in ? 1 : 0;
            my_rest += in ? 0 : len;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        my_in += __shfl_xor(my_in, off, 64);
        my_rest += __shfl_xor(my_rest, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_in[threadIdx.x >> 6] = my_in;
        s_rest[threadIdx.x >> 6] = my_rest;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int a = s_in[0] + s_in[1] + s_in[2] + s_in[3], bq = s_rest[0] + s_rest[1] + s_rest[2] + s_rest[3];
        if (a) atomicAdd(counters, a);
        if (bq) atomicAdd(counters + 1, bq);
    }
}

// The classes of the previous matrix still hold?  Per row: its word of the previous build against the new stream -- pattern id,
// length and every value bit for bit.  counters: [0] rows that match, [1] previously classed rows that do not (any: rebuild),
// [2] stream entries of the rows without a class.  (A driver that reassembles the same operator -- a time loop, the bench --
// pays one pass over the stream instead of the build: 2.9 -> 0.3 ms at 214^3 cells.)
__global__ __launch_bounds__(256) void k_cls_verify(const int32_t* __restrict__ rowptr, const double* __restrict__ val,
                                                    const uint16_t* __restrict__ pat, int32_t n, const uint32_t* __restrict__ cls,
                                                    const double* __restrict__ cls_val, const uint16_t* __restrict__ cls_pat,
                                                    int32_t n_cls, int32_t* __restrict__ counters, int cl) {
    __shared__ int32_t s_c[3][4];
    const int e = threadIdx.x & 7;
    int my_ok = 0, my_bad = 0, my_rest = 0;
    for (int64_t r64 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; r64 < n; r64 += ((int64_t)gridDim.x * blockDim.x) >> 3) {
        const int32_t r = (int32_t)r64;
        const int32_t b = rowptr[r], len = rowptr[r + 1] - b;
        const uint32_t w = cls[r];
        const int32_t id = (int32_t)(w >> 8);
        bool ok = w != CLS_NONE && id < n_cls && len <= cl && (uint32_t)pat[r] == (w & 255u) && cls_pat[id] == pat[r];
        long long diff = 0;
        if (ok)
            for (int j = e; j < len; j += 8) diff |= __double_as_longlong(val[b + j]) ^ __double_as_longlong(cls_val[(size_t)id * cl + j]);
        for (int off = 4; off > 0; off >>= 1) diff |= __shfl_xor(diff, off, 8);
        ok = ok && diff == 0;
        if (e == 0) {
            my_ok += ok ? 1 : 0;
            my_bad += (w != CLS_NONE && !ok) ? 1 : 0;
            my_rest += ok ? 0 : len;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        my_ok += __shfl_xor(my_ok, off, 64);
        my_bad += __shfl_xor(my_bad, off, 64);
        my_rest += __shfl_xor(my_rest, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_c[0][threadIdx.x >> 6] = my_ok;
        s_c[1][threadIdx.x >> 6] = my_bad;
        s_c[2][threadIdx.x >> 6] = my_rest;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int v = s_c[threadIdx.x][0] + s_c[threadIdx.x][1] + s_c[threadIdx.x][2] + s_c[threadIdx.x][3];
        if (v) atomicAdd(counters + threadIdx.x, v);
    }
}

// SpMV over row classes: a lane per row, RPT rows per lane 256 apart; a classed row reads its 4-byte (class, pattern) word, the
// column pattern (LDS) and the class's values (table), and x at row + offset (consecutive lanes = consecutive rows: coalesced); the other
// rows take their entries from the compacted stream.  Products and sums separate and in entry order: the bits of k_spmv_pat
// and k_spmv_win.
template <int RPT, int CL /* table stride = most entries of a classed row: 8, 16 or SPAT_L */>
__global__ __launch_bounds__(256) void k_spmv_cls(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                  const double* __restrict__ val, const uint16_t* __restrict__ pat,
                                                  const uint32_t* __restrict__ cls, const double* __restrict__ cls_val,
                                                  const int32_t* __restrict__ plen, const int32_t* __restrict__ pdelta, int32_t n_pat,
                                                  const double* __restrict__ x, double* __restrict__ y, int32_t n, SpmvEpi epi) {
    __shared__ int32_t sdelta[SPAT_P * SPAT_L];
    __shared__ int32_t slen[SPAT_P];
    const int tid = threadIdx.x;
    const int32_t nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int32_t lb = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
    const int32_t r0 = lb * (256 * RPT);
    uint32_t id[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int32_t r = r0 + tid + 256 * u;
        id[u] = r < n ? cls[r] : CLS_NONE;
    }
    for (int i = tid; i < n_pat * SPAT_L; i += 256) sdelta[i] = pdelta[i];
    for (int i = tid; i < n_pat; i += 256) slen[i] = plen[i];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int32_t r = r0 + tid + 256 * u;
        if (r >= n) continue;
        double s = 0.0;
        {
#pragma clang fp contract(off)
            // a wave whose rows are all classed with the SAME number of entries (the interior of a structured grid: 7 in 3D, 5 in
            // 2D) runs the row without a predicate -- the general form below tests every entry against the lane's length, and each
            // test compiles to an exec-mask branch around its load (214^3 cells, back to back: 70.0 -> 68.5 us, same bits: the
            // kernel is bound by the address path of its gathers, not by these instructions)
            const bool classed = id[u] != CLS_NONE;
            const int len_l = classed ? slen[id[u] & 255u] : -1;
            const int len_u = __builtin_amdgcn_readfirstlane(len_l);
            const bool uniform = CL == 8 && (len_u == 7 || len_u == 5) && __ballot(len_l == len_u) == __ballot(1);
            if (uniform) {      // (wave-uniform)
                const int32_t* __restrict__ dl = sdelta + (int)(id[u] & 255u) * SPAT_L;
                const vd2* __restrict__ cv = reinterpret_cast<const vd2*>(cls_val + (size_t)(id[u] >> 8) * CL);
                double av[8], xv[8];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const vd2 a2 = cv[j >> 1];
                    av[j] = a2.x;
                    av[j + 1] = a2.y;
                }
                if (len_u == 7) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) xv[j] = x[r + dl[j]];
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        const double pr = av[j] * xv[j];
                        s = s + pr;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 5; ++j) xv[j] = x[r + dl[j]];
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const double pr = av[j] * xv[j];
                        s = s + pr;
                    }
                }
            } else if (id[u] != CLS_NONE) {
                const int pid = (int)(id[u] & 255u);
                const int len = slen[pid];
                const int32_t* __restrict__ dl = sdelta + pid * SPAT_L;
                // (a class entry is CL doubles, 16-byte aligned, its padding zero: 16-byte loads, eight entries at a time)
                const vd2* __restrict__ cv = reinterpret_cast<const vd2*>(cls_val + (size_t)(id[u] >> 8) * CL);
#pragma unroll
                for (int j0 = 0; j0 < CL; j0 += 8) {
                    if (j0 < len) {
                        double xv[8], av[8];
#pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            const vd2 a2 = j0 + j < len ? cv[(j0 + j) >> 1] : vd2{0.0, 0.0};
                            av[j] = a2.x;
                            av[j + 1] = a2.y;
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) xv[j] = j0 + j < len ? x[r + dl[j0 + j]] : 0.0;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (j0 + j < len) {
                                const double pr = av[j] * xv[j];
                                s = s + pr;
                            }
                    }
                }
            } else {
                const int32_t b = rowptr[r], e = rowptr[r + 1];
                const uint16_t pid = pat[r];
                if (pid != SPAT_EXPL) {
                    const int32_t* __restrict__ dl = sdelta + (int)pid * SPAT_L;
                    for (int32_t p = b; p < e; ++p) {
                        const double pr = val[p] * x[r + dl[p - b]];
                        s = s + pr;
                    }
                } else {
                    for (int32_t p = b; p < e; ++p) {
                        const double pr = val[p] * x[colind[p]];
                        s = s + pr;
                    }
                }
            }
        }
        y[r] = epi_apply(epi, s, r);
    }
}

// Several ranks, option "halo_overlap": the rows that read ghost columns (a few node planes along the rank boundary) are
// listed at setup (k_bnd_flag -> scan -> k_bnd_list); the SpMV on row classes then runs over ALL rows while the ghost import
// travels on a second stream, and k_spmv_rows recomputes the listed rows when it has arrived -- the same entries in the same
// order (the stream's values are the class's values bit for bit), so y is the y of the un-overlapped product.
__global__ void k_bnd_flag(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int32_t n, int32_t* __restrict__ flag) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int f = 0;
    for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) f |= col[p] >= n ? 1 : 0;
    flag[r] = f;
}
__global__ void k_bnd_list(const int32_t* __restrict__ pos, int32_t n, int32_t* __restrict__ list) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && pos[r + 1] > pos[r]) list[pos[r]] = r;
}
__global__ void k_spmv_rows(const int32_t* __restrict__ list, int32_t nl, const int32_t* __restrict__ rowptr,
                            const int32_t* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x,
                            double* __restrict__ y, SpmvEpi epi) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nl) return;
    const int32_t r = list[i];
    double s = 0.0;
    {
#pragma clang fp contract(off)
        for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
            const double pr = val[p] * x[col[p]];
            s = s + pr;
        }
    }
    y[r] = epi_apply(epi, s, r);
}

// window -> first row table of the parity CSR (one-off per pattern)
static int spmv_window_rows(fedd_ctx* c) {
    const int32_t nb = (int32_t)(c->nnz / SP_CHUNK + 1);
    if (!c->spmv_rows_ready) {
        FEDD_TRY(c->d_spmv_rows.ensure((size_t)nb + 1));
        hipLaunchKernelGGL(k_spmv_block_rows, dim3((unsigned)((nb + 1 + 255) / 256)), dim3(256), 0, c->stream,
                           (const int32_t*)c->d_rowptr.p, (int32_t)c->n_rows, nb, c->d_spmv_rows.p);
        c->spmv_rows_ready = true;
    }
    return 0;
}

// (re)build the compacted stream after the matrix changed; part of the solver's first SpMV
static int spmv_compact_build(fedd_ctx* c) {
    const int32_t n = (int32_t)c->n_rows;
    const int32_t nb = (int32_t)(c->nnz / SP_CHUNK + 1);
    FEDD_TRY(spmv_window_rows(c));
    ScopedTimer ts(c, FEDD_T_SPMV_SETUP);
    FEDD_TRY(c->d_cs_wincnt.ensure((size_t)nb + 1));
    hipLaunchKernelGGL(k_cs_count, dim3((unsigned)nb), dim3(256), 0, c->stream, (const int32_t*)c->d_rowptr.p,
                       (const double*)c->d_val.p, (const int32_t*)c->d_spmv_rows.p, nb, c->spmv_drop_tol, c->d_cs_wincnt.p);
    int64_t total = 0;
    FEDD_TRY(exclusive_scan_i32(c, c->d_cs_wincnt.p, c->d_cs_wincnt.p, nb, &total));
    c->cs_nnz = total;
    FEDD_TRY(c->d_cs_rowptr.ensure((size_t)n + 1));
    FEDD_TRY(c->d_cs_col.ensure((size_t)total + 1));
    FEDD_TRY(c->d_cs_val.ensure((size_t)total + 4096 + 8));      // (k_spmv_pat reads whole 16-byte windows)
    hipLaunchKernelGGL(k_cs_fill, dim3((unsigned)nb), dim3(256), 0, c->stream, (const int32_t*)c->d_rowptr.p,
                       (const int32_t*)c->d_colind.p, (const double*)c->d_val.p, (const int32_t*)c->d_spmv_rows.p,
                       (const int32_t*)c->d_cs_wincnt.p, c->spmv_drop_tol, c->d_cs_rowptr.p, c->d_cs_col.p, c->d_cs_val.p);
    c->cs_tot32 = (int32_t)total;   // (lives in the context: the copy below is asynchronous)
    FEDD_HIP(hipMemcpyAsync(c->d_cs_rowptr.p + n, &c->cs_tot32, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    // (unlike k_spmv_pat the per-entry kernel gains nothing from one-trip windows: 214^3 cells, nu 8 / 7 / 6 / 5: 180 / 183 /
    // 184 / 184 us; 107^3: 24.2 -> 22.5 us with 6, 100^3: no difference -- 8 stays, the option is kept for A/B)
    c->cs_win_nu = (c->spmv_win_nu >= 4 && c->spmv_win_nu <= 8) ? c->spmv_win_nu : 8;
    const int32_t nbc = (int32_t)(total / (256 * c->cs_win_nu) + 1);
    FEDD_TRY(c->d_cs_rows.ensure((size_t)nbc + 1));
    hipLaunchKernelGGL(k_spmv_block_rows, dim3((unsigned)((nbc + 1 + 255) / 256)), dim3(256), 0, c->stream,
                       (const int32_t*)c->d_cs_rowptr.p, n, nbc, c->d_cs_rows.p, 256 * c->cs_win_nu);
    // column patterns (see k_spmv_pat)
    c->cs_npat = 0;
    c->cs_ncls = 0;
    // (matrices that fit the Infinity Cache keep the per-entry kernel -- measured: 100^3 cells 20.5 us against 25 us --,
    // so the dictionary is not built for them; option value 2 forces it)
    const bool big = 12.0 * (double)total > 256.0 * 1024.0 * 1024.0 || c->spmv_pattern == 2;
    // (rows longer than a pattern holds -- 16 entries; vector problems with full node blocks have 36 and more -- find none: the
    // three passes over the stream that would establish that cost 3.7 ms at cfg 5's share, so they are skipped by the average)
    const bool short_rows = total <= (int64_t)SPAT_L * std::max<int32_t>(n, 1);
    // (round 4: matrices in the Infinity Cache build the dictionary too when the row classes may pay -- they did on every
    // structured grid tried: 107^3 cells 23.2 -> 16.2 us --; without classes such a matrix goes back to the per-entry kernel)
    const bool try_classes = c->spmv_classes && !big && n >= 65536;
    if (c->spmv_pattern && total > 0 && (big || try_classes) && short_rows) {
        FEDD_TRY(c->d_cs_hash.ensure((size_t)n + SPAT_TS));
        FEDD_TRY(c->d_cs_pati.ensure((size_t)n + 2 * SPAT_TS + SPAT_P * (SPAT_L + 1) + 16));
        FEDD_TRY(c->d_cs_pat.ensure((size_t)n + 1));
        unsigned long long* hash = (unsigned long long*)c->d_cs_hash.p;
        unsigned long long* tkey = hash + n;
        int32_t* slot_of = c->d_cs_pati.p;
        int32_t* tmin = slot_of + n;
        int32_t* pat_of_slot = tmin + SPAT_TS;
        int32_t* plen = pat_of_slot + SPAT_TS;
        int32_t* pdelta = plen + SPAT_P;
        int32_t* counters = pdelta + SPAT_P * SPAT_L;      // claimed slots | patterns | explicit rows
        const dim3 b256(256), gr((unsigned)((n + 255) / 256)), gr8((unsigned)(((int64_t)n * 8 + 255) / 256));
        int32_t h[6] = {0, 0, 0, 0, 0, 0};    // claimed slots | patterns | explicit rows | longest row | longest pattern | rows that lost their pattern
        // the previous matrix' dictionary first: kept while every row that had a pattern still has it (one pass instead of four)
        bool pat_kept = false;
        if (c->spmv_keep_dict && c->cs_pat_tab_n == n && c->cs_pat_tab_npat > 0) {
            FEDD_HIP(hipMemsetAsync(counters + 2, 0, 4 * sizeof(int32_t), c->stream));
            hipLaunchKernelGGL(k_pat_rows, gr8, b256, 0, c->stream, (const int32_t*)c->d_cs_rowptr.p, (const int32_t*)c->d_cs_col.p, n,
                               (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)plen, (const int32_t*)pdelta,
                               c->d_cs_pat.p, counters + 2, (const uint16_t*)c->d_cs_pat.p, c->cs_pat_tab_npat);
            FEDD_HIP(hipMemcpyAsync(h + 2, counters + 2, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            FEDD_HIP(hipStreamSynchronize(c->stream));
            if (h[5] == 0 && h[2] == c->cs_pat_tab_nexpl) {
                pat_kept = true;
                h[1] = c->cs_pat_tab_npat;
                h[4] = c->cs_pat_tab_len;
            }
        }
        if (!pat_kept) {
        c->cs_pat_tab_n = -1;
        FEDD_HIP(hipMemsetAsync(tkey, 0, SPAT_TS * sizeof(unsigned long long), c->stream));
        FEDD_HIP(hipMemsetAsync(tmin, 0x7f, SPAT_TS * sizeof(int32_t), c->stream));
        FEDD_HIP(hipMemsetAsync(counters, 0, 8 * sizeof(int32_t), c->stream));
        hipLaunchKernelGGL(k_pat_hash, gr8, b256, 0, c->stream, (const int32_t*)c->d_cs_rowptr.p, (const int32_t*)c->d_cs_col.p, n, hash);
        // (a table that fills up means "no repeated patterns": the rows that find no slot keep explicit columns)
        hipLaunchKernelGGL(k_pat_insert, gr, b256, 0, c->stream, (const unsigned long long*)hash, n, tkey, tmin, slot_of, counters);
        hipLaunchKernelGGL(k_pat_table, dim3(1), b256, 0, c->stream, (const unsigned long long*)tkey, (const int32_t*)tmin,
                           (const int32_t*)c->d_cs_rowptr.p, (const int32_t*)c->d_cs_col.p, pat_of_slot, plen, pdelta, counters + 1);
        hipLaunchKernelGGL(k_pat_rows, gr8, b256, 0, c->stream, (const int32_t*)c->d_cs_rowptr.p, (const int32_t*)c->d_cs_col.p, n,
                           (const int32_t*)slot_of, (const int32_t*)pat_of_slot, (const int32_t*)plen, (const int32_t*)pdelta,
                           c->d_cs_pat.p, counters + 2, (const uint16_t*)nullptr, 0);
        FEDD_HIP(hipMemcpyAsync(h, counters, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        }
        // worth it when most rows found a pattern (an unstructured mesh fills the table and finds none)
        c->cs_max_len = std::max(1, h[3]);
        c->cs_npat = (h[1] >= 1 && (int64_t)h[2] * 4 <= (int64_t)n) ? std::min<int32_t>(h[1], SPAT_P) : 0;
        c->cs_nexpl = h[2];
        c->cs_pat_len = h[4];
        if (!pat_kept && c->cs_npat > 0) {      // (remembered for the next matrix)
            c->cs_pat_tab_n = n;
            c->cs_pat_tab_npat = c->cs_npat;
            c->cs_pat_tab_nexpl = h[2];
            c->cs_pat_tab_len = h[4];
        }
        if (c->cs_npat > 0) {
            c->cs_pat_nu = (c->spmv_pat_nu >= 2 && c->spmv_pat_nu <= 8)
                               ? c->spmv_pat_nu
                               // about 256 rows per window, one per lane and one trip: 512 nu values ~ 256 x the usual row length
                               // (214^3 cells, 6.8 entries per row: nu 3: 127 us, 4: 136, 5: 135, 6: 137)
                               : (int)std::min<int64_t>(8, std::max<int64_t>(2, total / (2 * (int64_t)std::max<int32_t>(n, 1))));
            const int32_t nbp = (int32_t)(total / (512 * c->cs_pat_nu) + 1);
            FEDD_TRY(c->d_cs_prows.ensure((size_t)nbp + 1));
            hipLaunchKernelGGL(k_spmv_block_rows, dim3((unsigned)((nbp + 1 + 255) / 256)), dim3(256), 0, c->stream,
                               (const int32_t*)c->d_cs_rowptr.p, n, nbp, c->d_cs_prows.p, 512 * c->cs_pat_nu);
        }
        // row classes on top of the column patterns (see k_cls_insert): rows that repeat their values bit for bit
        c->cs_ncls = 0;
        c->cs_cls_rows = 0;
        if (c->cs_npat > 0 && c->spmv_classes) {
            FEDD_TRY(c->d_cs_cls.ensure((size_t)n + 1));
            const int cl = c->cs_pat_len <= 8 ? 8 : (c->cs_pat_len <= 16 ? 16 : SPAT_L);       // table stride
            c->cs_cls_len = cl;
            FEDD_TRY(c->d_cs_clsval.ensure((size_t)CLS_MAX * cl));
            FEDD_TRY(c->d_cs_clspat.ensure((size_t)CLS_MAX));
            FEDD_TRY(c->d_cs_clsi.ensure((size_t)n + 2 * CLS_TS + 16));
            FEDD_TRY(c->d_cs_clskey.ensure((size_t)CLS_TS));
            unsigned long long* ckey = (unsigned long long*)c->d_cs_clskey.p;
            int32_t* cslot = c->d_cs_clsi.p;
            int32_t* cmin = cslot + n;
            int32_t* cof = cmin + CLS_TS;
            int32_t* ccnt = cof + CLS_TS;       // claimed | classes | classed rows | stream entries of the other rows
            // the previous matrix' classes first: if every row that had a class still matches it bit for bit, they are kept
            bool kept = false;
            if (c->spmv_keep_dict && c->cs_cls_tab_n == n && c->cs_cls_tab_len == cl && c->cs_cls_tab_ncls > 0 && c->cs_cls_tab_rows > 0) {
                FEDD_HIP(hipMemsetAsync(ccnt, 0, 8 * sizeof(int32_t), c->stream));
                hipLaunchKernelGGL(k_cls_verify, dim3(std::min<unsigned>(gr8.x, 8192u)), b256, 0, c->stream, (const int32_t*)c->d_cs_rowptr.p, (const double*)c->d_cs_val.p,
                                   (const uint16_t*)c->d_cs_pat.p, n, (const uint32_t*)c->d_cs_cls.p, (const double*)c->d_cs_clsval.p,
                                   (const uint16_t*)c->d_cs_clspat.p, c->cs_cls_tab_ncls, ccnt, cl);
                int32_t hv[3] = {0, 0, 0};
                FEDD_HIP(hipMemcpyAsync(hv, ccnt, sizeof(hv), hipMemcpyDeviceToHost, c->stream));
                FEDD_HIP(hipStreamSynchronize(c->stream));
                if (hv[1] == 0 && hv[0] == c->cs_cls_tab_rows) {
                    kept = true;
                    c->cs_ncls = c->cs_cls_tab_ncls;
                    c->cs_cls_rows = hv[0];
                    c->cs_cls_rest = hv[2];
                }
            }
            if (!kept) {
            c->cs_cls_tab_ncls = 0;
            FEDD_HIP(hipMemsetAsync(ckey, 0, CLS_TS * sizeof(unsigned long long), c->stream));
            FEDD_HIP(hipMemsetAsync(cmin, 0x7f, CLS_TS * sizeof(int32_t), c->stream));
            FEDD_HIP(hipMemsetAsync(ccnt, 0, 8 * sizeof(int32_t), c->stream));
            hipLaunchKernelGGL(k_cls_insert, gr8, b256, 0, c->stream, (const int32_t*)c->d_cs_rowptr.p, (const double*)c->d_cs_val.p,
                               (const uint16_t*)c->d_cs_pat.p, n, ckey, cmin, cslot, ccnt);
            hipLaunchKernelGGL(k_cls_table, dim3(1), dim3(1024), 0, c->stream, (const unsigned long long*)ckey, cof, ccnt + 1);
            hipLaunchKernelGGL(k_cls_fill, dim3(CLS_TS / 256), b256, 0, c->stream, (const int32_t*)cof, (const int32_t*)cmin,
                               (const int32_t*)c->d_cs_rowptr.p, (const double*)c->d_cs_val.p, (const uint16_t*)c->d_cs_pat.p,
                               c->d_cs_clsval.p, c->d_cs_clspat.p, cl);
            hipLaunchKernelGGL(k_cls_rows, dim3(std::min<unsigned>(gr8.x, 8192u)), b256, 0, c->stream, (const int32_t*)c->d_cs_rowptr.p, (const double*)c->d_cs_val.p,
                               (const uint16_t*)c->d_cs_pat.p, n, (const int32_t*)cslot, (const int32_t*)cof,
                               (const double*)c->d_cs_clsval.p, (const uint16_t*)c->d_cs_clspat.p, c->d_cs_cls.p, ccnt + 2, cl);
            int32_t hc[4] = {0, 0, 0, 0};
            FEDD_HIP(hipMemcpyAsync(hc, ccnt, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
            FEDD_HIP(hipStreamSynchronize(c->stream));
            if (getenv("FEDD_SPMV_DEBUG")) fprintf(stderr, "[spmv classes] n %d: slots claimed %d, classes %d, rows in classes %d, stream entries outside %d\n", n, hc[0], hc[1], hc[2], hc[3]);
            // worth it when nearly every row is in a class (the other rows go through a per-row loop)
            if (hc[1] >= 1 && (int64_t)hc[2] * 100 >= (int64_t)n * c->spmv_cls_cover) {
                c->cs_ncls = std::min<int32_t>(hc[1], CLS_MAX);
                c->cs_cls_rows = hc[2];
                c->cs_cls_rest = hc[3];
                c->cs_cls_tab_n = n;
                c->cs_cls_tab_len = cl;
                c->cs_cls_tab_ncls = c->cs_ncls;
                c->cs_cls_tab_rows = hc[2];
            }
            }   // (!kept)
        }
        // (the dictionary was only tried for the classes' sake: a matrix in the Infinity Cache, or rows longer than k_spmv_pat unrolls)
        if ((!big || c->cs_pat_len > SPAT_LK) && c->cs_ncls == 0) c->cs_npat = 0;
    }
    // rows that read ghost columns (several ranks; for the overlapped import of the class SpMV, see k_bnd_flag)
    c->cs_nbnd = -1;
    if (c->cs_ncls > 0 && c->n_cols > c->n_rows && total > 0) {
        FEDD_TRY(c->d_cs_bnd.ensure((size_t)2 * n + 4));
        int32_t* bflag = c->d_cs_bnd.p;             // [n + 1] flags -> positions
        int32_t* blist = bflag + n + 2;             // [<= n] the rows
        const dim3 gq((unsigned)((n + 255) / 256));
        hipLaunchKernelGGL(k_bnd_flag, gq, dim3(256), 0, c->stream, (const int32_t*)c->d_cs_rowptr.p, (const int32_t*)c->d_cs_col.p, n, bflag);
        int64_t nb_rows = 0;
        FEDD_TRY(exclusive_scan_i32(c, bflag, bflag, n, &nb_rows));
        hipLaunchKernelGGL(k_bnd_list, gq, dim3(256), 0, c->stream, (const int32_t*)bflag, n, blist);
        c->cs_nbnd = (int32_t)nb_rows;
    }
    // 16-bit columns for the per-entry window kernel (option "spmv_col16"; decided by the data: every window's column span)
    // (not for a stream that goes through the column patterns: k_spmv_pat reads no column of a row that has one)
    c->cs_col16 = false;
    if (c->spmv_col16 && total > 0 && c->cs_npat == 0) {
        FEDD_TRY(c->d_cs_col16.ensure((size_t)total + 4096 + 8));
        FEDD_TRY(c->d_cs_wbase.ensure((size_t)nbc + 1));
        FEDD_TRY(c->d_flags.ensure(16));
        int32_t* flag = c->d_flags.p + 7;
        FEDD_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), c->stream));
        hipLaunchKernelGGL(k_cs_col16, dim3((unsigned)nbc), dim3(256), 0, c->stream, (const int32_t*)c->d_cs_col.p, (int32_t)total,
                           256 * c->cs_win_nu, c->d_cs_wbase.p, c->d_cs_col16.p, flag);
        c->cs_col16 = true;     // (per window: nothing for the host to wait for; fedd_spmv_col_bytes reads the count when asked)
    }
    ts.stop();
    FEDD_HIP(hipGetLastError());
    c->cs_valid = true;
    return 0;
}

__global__ void k_epi_only(double* __restrict__ y, SpmvEpi epi, int32_t n) {   // y = 0 * x - theta * sub (matrix without entries)
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = epi_apply(epi, 0.0, i);
}

int spmv_owned(fedd_ctx* c, const double* d_x_owned, double* d_y_owned, bool x_has_tail, const double* d_sub, double theta,
               int use_compact) {
    const double* x = d_x_owned;
    const SpmvEpi epi{d_sub, theta};
    const bool compact = use_compact < 0 ? c->spmv_compact != 0 : use_compact != 0;
    // (the solver's stream is built before the ghost import is issued: the overlapped import below needs its row list)
    if (compact && c->spmv_kind == 0 && c->max_row_nnz <= 256 && c->nnz > 0 && !c->cs_valid) FEDD_TRY(spmv_compact_build(c));
    bool overlapped = false;
    if (c->n_cols != c->n_rows || !c->halo.peers.empty()) {   // also a rank that only sends takes part
        double* xb = const_cast<double*>(d_x_owned);
        if (!x_has_tail) {   // (else the caller's buffer takes the ghost values behind its owned entries)
            FEDD_HIP(hipMemcpyAsync(c->d_xcol.p, d_x_owned, (size_t)c->n_rows * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            xb = c->d_xcol.p;
            x = c->d_xcol.p;
        }
        // option "halo_overlap" with the class kernel (its build is current and lists the rows that read ghost columns): the
        // import goes to a second stream, the product over all rows runs meanwhile, the listed rows are redone afterwards
        const bool cls_path = compact && c->spmv_kind == 0 && c->max_row_nnz <= 256 && c->nnz > 0 && c->cs_valid && c->cs_npat > 0 &&
                              c->spmv_pattern && c->cs_ncls > 0 && c->spmv_classes && c->cs_nnz > 0;
        if (c->halo_overlap && c->nranks > 1 && cls_path && c->cs_nbnd >= 0) {
            if (!c->stream2) {
                FEDD_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
                FEDD_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                FEDD_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
            hipStream_t main_stream = c->stream;
            FEDD_HIP(hipEventRecord(c->ev_fork, main_stream));       // the owned part of x is complete here
            FEDD_HIP(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            c->stream = c->stream2;
            const int rc = halo_import(c, xb, c->dofs);
            c->stream = main_stream;
            if (rc) return rc;
            FEDD_HIP(hipEventRecord(c->ev_join, c->stream2));
            overlapped = true;
        } else {
            FEDD_TRY(halo_import(c, xb, c->dofs));
        }
    }
    const double avg = c->n_rows ? (double)c->nnz / (double)c->n_rows : 1.0;
    const int32_t n = (int32_t)c->n_rows;
    // spmv_kind: 0 = automatic (CSR-window, CSR-stream for rows longer than 256, row-per-lane-group for
    // very long rows), 1 = row-per-lane-group, 2 = CSR-stream
    const bool windowed = c->spmv_kind == 0 && c->max_row_nnz <= 256 && c->nnz > 0;
    const bool streamed = !windowed && (c->spmv_kind == 0 || c->spmv_kind == 2) && avg <= 64.0 && c->max_row_nnz <= 2048;
    if (windowed && compact) {
        // the compacted stream (numerically zero entries left out): same kernel, fewer bytes
        if (!c->cs_valid) FEDD_TRY(spmv_compact_build(c));
        const int wnu = c->cs_win_nu;
        const int32_t nbc = (int32_t)(c->cs_nnz / (256 * wnu) + 1);
        const int32_t ovh = (int32_t)std::max<int64_t>(c->max_row_nnz, 1);
        const size_t lds = (size_t)(256 * wnu + ovh) * sizeof(double);
        const bool nt = c->spmv_nt < 0 ? 12.0 * (double)c->cs_nnz > 256.0 * 1024.0 * 1024.0 : c->spmv_nt != 0;
        ScopedTimer ts(c, FEDD_T_SPMV);
        if (c->cs_nnz == 0) {
            if (epi.sub) hipLaunchKernelGGL(k_epi_only, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_y_owned, epi, n);
            else FEDD_HIP(hipMemsetAsync(d_y_owned, 0, (size_t)n * sizeof(double), c->stream));
        } else if (c->cs_npat > 0 && c->spmv_pattern && c->cs_ncls > 0 && c->spmv_classes) {
            const int32_t* plen = c->d_cs_pati.p + n + 2 * SPAT_TS;
            const int32_t* pdelta = plen + SPAT_P;
#define SPMV_CLS(RPT_, CL_)                                                                                                     \
    hipLaunchKernelGGL((k_spmv_cls<RPT_, CL_>), dim3((unsigned)((n + 256 * RPT_ - 1) / (256 * RPT_))), dim3(256), 0, c->stream,     \
                       (const int32_t*)c->d_cs_rowptr.p, (const int32_t*)c->d_cs_col.p, (const double*)c->d_cs_val.p,                 \
                       (const uint16_t*)c->d_cs_pat.p, (const uint32_t*)c->d_cs_cls.p, (const double*)c->d_cs_clsval.p, plen, pdelta, \
                       c->cs_npat, x, d_y_owned, n, epi)
            if (c->cs_cls_len <= 8) SPMV_CLS(4, 8);
            else if (c->cs_cls_len <= 16) SPMV_CLS(4, 16);
            else SPMV_CLS(2, SPAT_L);
#undef SPMV_CLS
            if (overlapped) {   // the ghost values have arrived: the rows that read them, again
                FEDD_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
                overlapped = false;
                if (c->cs_nbnd > 0)
                    hipLaunchKernelGGL(k_spmv_rows, dim3((unsigned)((c->cs_nbnd + 255) / 256)), dim3(256), 0, c->stream,
                                       (const int32_t*)(c->d_cs_bnd.p + n + 2), c->cs_nbnd, (const int32_t*)c->d_cs_rowptr.p,
                                       (const int32_t*)c->d_cs_col.p, (const double*)c->d_cs_val.p, x, d_y_owned, epi);
            }
        } else if (c->cs_npat > 0 && c->spmv_pattern) {
            const int32_t* plen = c->d_cs_pati.p + n + 2 * SPAT_TS;
            const int32_t* pdelta = plen + SPAT_P;
            const int nu = c->cs_pat_nu;                        // 16-byte loads per lane: window = 512 nu values
            const int32_t nbp = (int32_t)(c->cs_nnz / (512 * nu) + 1);
            const size_t ldp = (size_t)(512 * nu + ovh) * sizeof(double);
#define SPMV_PAT1(NT_, NU_, LU_)                                                                                                    \
    hipLaunchKernelGGL((k_spmv_pat<NT_, NU_, LU_>), dim3((unsigned)nbp), dim3(256), ldp, c->stream, (const int32_t*)c->d_cs_rowptr.p, \
                       (const int32_t*)c->d_cs_col.p, (const double*)c->d_cs_val.p, (const uint16_t*)c->d_cs_pat.p, plen,           \
                       pdelta, c->cs_npat, x, d_y_owned, (const int32_t*)c->d_cs_prows.p, nbp, (int32_t)c->cs_nnz, ovh, epi)
#define SPMV_PAT(NT_, NU_)                                 \
    if (c->cs_pat_len <= 8) SPMV_PAT1(NT_, NU_, 8);         \
    else SPMV_PAT1(NT_, NU_, SPAT_LK)
#define SPMV_PAT_NU(NT_)                 \
    switch (nu) {                        \
        case 2: SPMV_PAT(NT_, 2); break; \
        case 3: SPMV_PAT(NT_, 3); break; \
        case 4: SPMV_PAT(NT_, 4); break; \
        case 5: SPMV_PAT(NT_, 5); break; \
        case 6: SPMV_PAT(NT_, 6); break; \
        case 7: SPMV_PAT(NT_, 7); break; \
        default: SPMV_PAT(NT_, 8); break; \
    }
            if (nt) { SPMV_PAT_NU(true) } else { SPMV_PAT_NU(false) }
#undef SPMV_PAT_NU
#undef SPMV_PAT
#undef SPMV_PAT1
        } else {
#define SPMV_WIN(NT_, NU_)                                                                                                       \
    if (c->cs_col16)                                                                                                             \
        hipLaunchKernelGGL((k_spmv_win<NT_, NU_, true>), dim3((unsigned)nbc), dim3(256), lds, c->stream,                         \
                           (const int32_t*)c->d_cs_rowptr.p, (const int32_t*)c->d_cs_col.p, (const double*)c->d_cs_val.p, x,     \
                           d_y_owned, (const int32_t*)c->d_cs_rows.p, nbc, (int32_t)c->cs_nnz, ovh, epi,                         \
                           (const uint16_t*)c->d_cs_col16.p, (const int32_t*)c->d_cs_wbase.p);                                   \
    else                                                                                                                         \
        hipLaunchKernelGGL((k_spmv_win<NT_, NU_>), dim3((unsigned)nbc), dim3(256), lds, c->stream, (const int32_t*)c->d_cs_rowptr.p, \
                           (const int32_t*)c->d_cs_col.p, (const double*)c->d_cs_val.p, x, d_y_owned,                            \
                           (const int32_t*)c->d_cs_rows.p, nbc, (int32_t)c->cs_nnz, ovh, epi, (const uint16_t*)nullptr,           \
                           (const int32_t*)nullptr)
#define SPMV_WIN_NU(NT_)                 \
    switch (wnu) {                       \
        case 4: SPMV_WIN(NT_, 4); break; \
        case 5: SPMV_WIN(NT_, 5); break; \
        case 6: SPMV_WIN(NT_, 6); break; \
        case 7: SPMV_WIN(NT_, 7); break; \
        default: SPMV_WIN(NT_, 8); break; \
    }
            if (nt) { SPMV_WIN_NU(true) } else { SPMV_WIN_NU(false) }
#undef SPMV_WIN_NU
#undef SPMV_WIN
        }
        ts.stop();
        FEDD_HIP(hipGetLastError());
        return 0;
    }
    if (windowed || streamed) {
        const int32_t nb = (int32_t)(c->nnz / SP_CHUNK + 1);
        FEDD_TRY(spmv_window_rows(c));
        const int32_t ovh = (int32_t)std::max<int64_t>(c->max_row_nnz, 1);
        const size_t lds = (size_t)(SP_CHUNK + ovh) * sizeof(double);
        ScopedTimer ts(c, FEDD_T_SPMV);
        // non-temporal matrix stream: by default for matrices that do not fit the 256 MB Infinity Cache anyway
        const bool nt = c->spmv_nt < 0 ? 12.0 * (double)c->nnz > 256.0 * 1024.0 * 1024.0 : c->spmv_nt != 0;
        if (windowed && nt)
            hipLaunchKernelGGL(k_spmv_win<true>, dim3((unsigned)nb), dim3(256), lds, c->stream, (const int32_t*)c->d_rowptr.p,
                               (const int32_t*)c->d_colind.p, (const double*)c->d_val.p, x, d_y_owned,
                               (const int32_t*)c->d_spmv_rows.p, nb, (int32_t)c->nnz, ovh, epi);
        else if (windowed)
            hipLaunchKernelGGL(k_spmv_win<false>, dim3((unsigned)nb), dim3(256), lds, c->stream, (const int32_t*)c->d_rowptr.p,
                               (const int32_t*)c->d_colind.p, (const double*)c->d_val.p, x, d_y_owned,
                               (const int32_t*)c->d_spmv_rows.p, nb, (int32_t)c->nnz, ovh, epi);
        else
            hipLaunchKernelGGL(k_spmv_stream, dim3((unsigned)nb), dim3(256), lds, c->stream, (const int32_t*)c->d_rowptr.p,
                               (const int32_t*)c->d_colind.p, (const double*)c->d_val.p, x, d_y_owned,
                               (const int32_t*)c->d_spmv_rows.p, nb, epi);
        ts.stop();
        FEDD_HIP(hipGetLastError());
        return 0;
    }
    ScopedTimer t(c, FEDD_T_SPMV);
#define SPMV_LAUNCH(L)                                                                                        \
    hipLaunchKernelGGL(k_spmv<L>, dim3((unsigned)(((int64_t)n * L + 255) / 256)), dim3(256), 0, c->stream,      \
                       (const int32_t*)c->d_rowptr.p, (const int32_t*)c->d_colind.p, (const double*)c->d_val.p, \
                       x, d_y_owned, n, epi)
    if (avg <= 4.0) SPMV_LAUNCH(4);
    else if (avg <= 10.0) SPMV_LAUNCH(8);
    else if (avg <= 24.0) SPMV_LAUNCH(16);
    else if (avg <= 48.0) SPMV_LAUNCH(32);
    else SPMV_LAUNCH(64);
#undef SPMV_LAUNCH
    t.stop();
    FEDD_HIP(hipGetLastError());
    return 0;
}

}  // namespace fedd
