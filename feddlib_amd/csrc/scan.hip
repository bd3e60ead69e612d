// Device-wide exclusive scan and max-reduction (hand-written; used by the symbolic phase and the
// Schwarz setup).  Three-level block scan: 256 threads x 8 items per block.
#include "fedd_internal.hpp"
#include <algorithm>
#include <climits>

namespace fedd {

namespace {
constexpr int ST = 256, SI = 8, SB = ST * SI;

// logical input: in[i] for i < n_in, 0 beyond; writes out[i] for i < n_out, block totals to bsum.
// in == out is allowed (each thread reads its items before any thread writes them).
template <class TI, class TO>
__global__ __launch_bounds__(ST) void k_scan_block(const TI* __restrict__ in, TO* out, int64_t* __restrict__ bsum,
                                                   int64_t n_in, int64_t n_out) {
    __shared__ int64_t sh[ST];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * SB + (int64_t)tid * SI;
    int64_t v[SI];
    int64_t tot = 0;
#pragma unroll
    for (int k = 0; k < SI; ++k) {
        const int64_t i = base + k;
        v[k] = i < n_in ? (int64_t)in[i] : 0;
        tot += v[k];
    }
    sh[tid] = tot;
    __syncthreads();
    for (int off = 1; off < ST; off <<= 1) {
        const int64_t t = tid >= off ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    int64_t run = sh[tid] - tot;
#pragma unroll
    for (int k = 0; k < SI; ++k) {
        const int64_t i = base + k;
        if (i < n_out) out[i] = (TO)run;
        run += v[k];
    }
    if (tid == ST - 1) bsum[blockIdx.x] = sh[tid];
}

template <class TO>
__global__ __launch_bounds__(ST) void k_scan_add(TO* out, const int64_t* __restrict__ boff, int64_t n_out) {
    const int64_t off = boff[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SB;
    for (int k = threadIdx.x; k < SB; k += ST) {
        const int64_t i = base + k;
        if (i < n_out) out[i] = (TO)((int64_t)out[i] + off);
    }
}

template <class TI, class TO>
int scan_level(fedd_ctx* c, const TI* in, TO* out, int64_t n_in, int level) {
    const int64_t n_out = n_in + 1;
    const int64_t nb = (n_out + SB - 1) / SB;
    FEDD_CHECK(level < 3, "device scan: input too large");
    FEDD_TRY(c->d_scan[level].ensure((size_t)nb + 1));
    int64_t* bs = c->d_scan[level].p;
    hipLaunchKernelGGL((k_scan_block<TI, TO>), dim3((unsigned)nb), dim3(ST), 0, c->stream, in, out, bs, n_in, n_out);
    if (nb > 1) {
        FEDD_TRY((scan_level<int64_t, int64_t>(c, bs, bs, nb, level + 1)));
        hipLaunchKernelGGL((k_scan_add<TO>), dim3((unsigned)nb), dim3(ST), 0, c->stream, out, bs, n_out);
    }
    FEDD_HIP(hipGetLastError());
    return 0;
}

__global__ void k_max_i32(const int32_t* __restrict__ in, int64_t n, int32_t* out) {
    __shared__ int32_t sh[256];
    int32_t m = INT32_MIN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = max(m, in[i]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = max(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(out, sh[0]);
}
}  // namespace

// out has n+1 entries; out[n] = total.  total_out (nullable) triggers one D2H copy + sync.
int exclusive_scan_i32(fedd_ctx* c, const int32_t* d_in, int32_t* d_out, int64_t n, int64_t* total_out) {
    FEDD_TRY((scan_level<int32_t, int32_t>(c, d_in, d_out, n, 0)));
    if (total_out) {
        int32_t t = 0;
        FEDD_HIP(hipMemcpyAsync(&t, d_out + n, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        *total_out = t;
    }
    return 0;
}

int exclusive_scan_i64(fedd_ctx* c, const int64_t* d_in, int64_t* d_out, int64_t n, int64_t* total_out) {
    FEDD_TRY((scan_level<int64_t, int64_t>(c, d_in, d_out, n, 0)));
    if (total_out) {
        int64_t t = 0;
        FEDD_HIP(hipMemcpyAsync(&t, d_out + n, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        *total_out = t;
    }
    return 0;
}

int reduce_max_i32(fedd_ctx* c, const int32_t* d_in, int64_t n, int32_t* out) {
    FEDD_TRY(c->d_flags.ensure(16));
    int32_t* d = c->d_flags.p;
    const int32_t init = INT32_MIN;
    FEDD_HIP(hipMemcpyAsync(d, &init, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    const int nb = (int)std::min<int64_t>(1024, (n + 255) / 256);
    if (n > 0) hipLaunchKernelGGL(k_max_i32, dim3(nb), dim3(256), 0, c->stream, d_in, n, d);
    FEDD_HIP(hipMemcpyAsync(out, d, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace fedd
