// Read-only streaming bandwidth ceiling on this box (calibration for the roofline numbers):
// sums `bytes` of f64 with 16-byte loads, several launch shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ p, size_t n2, double* out, int unroll_dummy) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    double s = 0.0;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        const double2 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        s += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    for (; i < n2; i += stride) s += p[i].x + p[i].y;
    if (s == 12345.678) out[0] = s;
}

// block-contiguous: each workgroup streams its own contiguous 24 KB (the Schwarz apply pattern)
__global__ __launch_bounds__(256) void k_read_blocks(const double2* __restrict__ p, size_t per_block2, double* out) {
    const double2* q = p + (size_t)blockIdx.x * per_block2;
    double s = 0.0;
    for (size_t i = threadIdx.x; i < per_block2; i += 256) s += q[i].x + q[i].y;
    if (s == 12345.678) out[0] = s;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 800) << 20;   // MiB streamed per launch
    printf("footprint %zu MiB\n", bytes >> 20);
    double* d;
    double* out;
    hipMalloc(&d, bytes);
    hipMalloc(&out, 8);
    hipMemset(d, 0, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const size_t n2 = bytes / 16;
    for (int grid : {2048, 8192, 32768}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const double2*)d, n2, out, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("grid-stride grid %d: %.1f GB/s\n", grid, bytes * 10 / (ms * 1e6));
        }
    }
    for (size_t per : {(size_t)1536, (size_t)6144}) {  // 24 KB and 96 KB per workgroup
        const int grid = (int)(n2 / per);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(k_read_blocks, dim3(grid), dim3(256), 0, 0, (const double2*)d, per, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("block-contiguous %zu KB x %d: %.1f GB/s\n", per * 16 / 1024, grid, (double)grid * per * 16 * 10 / (ms * 1e6));
        }
    }
    return 0;
}
