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

// ------------------------------------------------------------------------------------------------
// Stable LSD radix sort of (key, value) pairs of int32 (non-negative keys), 8 bits per pass, hand-written (rounds 1-3 called
// rocPRIM through hipCUB here: the one non-native dependency of the path).  A pass: k_rs_count (a 256-bin histogram per
// workgroup tile of RS_TILE items, LDS atomics) -> device scan of the bin-major table [bin][tile] -> k_rs_scatter (the tile
// again, 256 items at a time in order; within a round the lanes of a wave that carry the same digit find each other with eight
// ballots, the lowest of them takes the digit's running offset from LDS, and the four waves go one after the other: ranks
// follow the input order, so the sort is stable).  Used by the Schwarz setup (subdomains by representative) and the coarse
// levels (nodes by lattice cell).
// ------------------------------------------------------------------------------------------------
constexpr int RS_ITEMS = 8, RS_TILE = 256 * RS_ITEMS;

__global__ __launch_bounds__(256) void k_rs_count(const int32_t* __restrict__ keys, int32_t n, int shift, int32_t ntile,
                                                  int32_t* __restrict__ hist /* [256][ntile] */) {
    __shared__ int32_t h[256];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int u = 0; u < RS_ITEMS; ++u) {
        const int64_t i = base + u * 256 + tid;
        if (i < n) atomicAdd(&h[((uint32_t)keys[i] >> shift) & 255u], 1);
    }
    __syncthreads();
    hist[(int64_t)tid * ntile + blockIdx.x] = h[tid];
}

__global__ __launch_bounds__(256) void k_rs_scatter(const int32_t* __restrict__ keys, const int32_t* __restrict__ vals, int32_t n,
                                                    int shift, int32_t ntile, const int32_t* __restrict__ offs /* scanned hist */,
                                                    int32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out) {
    __shared__ int32_t run[256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    run[tid] = offs[(int64_t)tid * ntile + blockIdx.x];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    for (int u = 0; u < RS_ITEMS; ++u) {
        const int64_t i = base + u * 256 + tid;
        const bool on = i < n;
        const int32_t k = on ? keys[i] : 0, v = on ? vals[i] : 0;
        const uint32_t d = ((uint32_t)k >> shift) & 255u;
        // lanes of this wave with the same digit (and in range)
        uint64_t same = __ballot(on);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const int rank = __builtin_popcountll(same & ((1ull << lane) - 1ull));
        const int cnt = __builtin_popcountll(same);
        const int leader = on ? __builtin_ctzll(same) : 0;
        int32_t pos = 0;
        for (int ww = 0; ww < 4; ++ww) {        // the waves in order: stable
            if (w == ww && on) {
                int32_t b0 = 0;
                if (lane == leader) {
                    b0 = run[d];
                    run[d] = b0 + cnt;
                }
                pos = __shfl(b0, leader, 64) + rank;
            }
            __syncthreads();
        }
        if (on) {
            keys_out[pos] = k;
            vals_out[pos] = v;
        }
    }
}

// keys[0] / vals[0] hold the input; the passes ping-pong between buffers 0 and 1; *cur_out = the buffer that holds the result.
// Bits [0, bits) of the keys take part.
int radix_sort_pairs_i32(fedd_ctx* c, int32_t* keys[2], int32_t* vals[2], int32_t n, int bits, int* cur_out) {
    *cur_out = 0;
    if (n <= 1 || bits <= 0) return 0;
    const int32_t ntile = (int32_t)(((int64_t)n + RS_TILE - 1) / RS_TILE);
    FEDD_TRY(c->d_rs_hist.ensure((size_t)256 * ntile + 2));
    int cur = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(k_rs_count, dim3((unsigned)ntile), dim3(256), 0, c->stream, (const int32_t*)keys[cur], n, shift, ntile,
                           c->d_rs_hist.p);
        FEDD_TRY(exclusive_scan_i32(c, c->d_rs_hist.p, c->d_rs_hist.p, (int64_t)256 * ntile, nullptr));
        hipLaunchKernelGGL(k_rs_scatter, dim3((unsigned)ntile), dim3(256), 0, c->stream, (const int32_t*)keys[cur],
                           (const int32_t*)vals[cur], n, shift, ntile, (const int32_t*)c->d_rs_hist.p, keys[cur ^ 1], vals[cur ^ 1]);
        cur ^= 1;
    }
    FEDD_HIP(hipGetLastError());
    *cur_out = cur;
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
