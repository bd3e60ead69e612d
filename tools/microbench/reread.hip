// Does a second read of a workgroup's row block, right after the first, come back faster than HBM (L2 / Infinity Cache)?
// Layout of the Krylov basis: K columns of n doubles (column-major, ld = n); a workgroup takes 512 rows (2 per lane) of all K columns --
// the access pattern of the block Gram-Schmidt sweeps (gmres.hip k_blockaxpy / k_blockdot).  once: one pass over the columns;
// twice: two passes over the same rows (what a fused update + dot sweep would do); the gain of fusing = 2 x once - twice.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int PASSES, bool NT>
__global__ __launch_bounds__(256) void k_rows(const double* __restrict__ V, size_t n, int K, double* out) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const size_t r = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (r + 1 >= n) return;
    double s = 0.0;
    for (int p = 0; p < PASSES; ++p) {
        int c = 0;
        for (; c + 8 <= K; c += 8) {
            v2d q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const v2d* a = reinterpret_cast<const v2d*>(V + (size_t)(c + u) * n + r);
                q[u] = (NT && p == PASSES - 1) ? __builtin_nontemporal_load(a) : *a;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += q[u].x * (p + 1) + q[u].y;
        }
    }
    if (s == 12345.678) out[0] = s;
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? atol(argv[1]) : 9938376;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    double* out;
    hipMalloc(&out, 8);
    for (int K : {32, 64, 96}) {
        double* d;
        hipMalloc(&d, n * K * 8);
        hipMemset(d, 0, n * K * 8);
        const int grid = (int)((n / 2 + 255) / 256);
        float ms1 = 0, ms2 = 0, ms2nt = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((k_rows<1, false>), dim3(grid), dim3(256), 0, 0, (const double*)d, n, K, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms1, a, b);
            hipEventRecord(a);
            for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((k_rows<2, false>), dim3(grid), dim3(256), 0, 0, (const double*)d, n, K, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms2, a, b);
            hipEventRecord(a);
            for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((k_rows<2, true>), dim3(grid), dim3(256), 0, 0, (const double*)d, n, K, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms2nt, a, b);
        }
        const double gb = (double)n * K * 8 / 1e9;
        printf("K %3d (%.2f GB): one pass %.3f ms (%.0f GB/s), two passes over the same rows %.3f ms (second pass non-temporal: %.3f ms); two separate sweeps %.3f ms\n",
               K, gb, ms1 / 5, gb / (ms1 / 5) * 1e3, ms2 / 5, ms2nt / 5, 2 * ms1 / 5);
        hipFree(d);
    }
    return 0;
}
