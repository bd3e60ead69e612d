// What the f64 matrix cores of this box sustain (calibration of the Schwarz apply's roof): bare loops of
// v_mfma_f64_16x16x4_f64 on register operands, NACC independent accumulators per wave, one workgroup of four waves per
// SIMD slot; with s_memtime around the loop to separate cycles per instruction from the clock the chip holds.
//   hipcc --offload-arch=gfx950 -O3 -o bin/mfma_f64 mfma_f64.hip && bin/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

// RANDOM: operands with random mantissas (what an inverse and a residual look like) instead of a few small multiples --
// the cycles per instruction are the same, the clock the chip holds under the load is not
__device__ inline double rnd_double(unsigned long long& st) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(long long)(st >> 11) * (1.0 / 9007199254740992.0) - 0.25;
}

template <int NACC, bool RANDOM>
__global__ __launch_bounds__(256) void k_mfma(double* out, long long* cyc, int iters, double seed) {
    d4 acc[NACC];
    double a[NACC], b[4];
    unsigned long long st = (unsigned long long)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + (unsigned long long)seed;
    for (int i = 0; i < NACC; ++i) {
        acc[i] = d4{0.0, 0.0, 0.0, 0.0};
        a[i] = RANDOM ? rnd_double(st) : seed * (threadIdx.x % 7 + i + 1) * 1e-3;
    }
    for (int i = 0; i < 4; ++i) b[i] = RANDOM ? rnd_double(st) : seed * (threadIdx.x % 5 + i + 1) * 1e-3;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[k], acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345e300) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// the Schwarz apply's product phase alone: RT x KW fragments of A in registers, KW fragments of B renewed per batch (from a
// register rotation, no memory), RT accumulators from zero per batch, the K-split partial tile folded into a sink
template <int RT, int KW, bool RANDOM>
__global__ __launch_bounds__(256, 2) void k_tile(double* out, long long* cyc, int batches, double seed) {
    double a[RT][KW], b[KW];
    unsigned long long st = (unsigned long long)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + (unsigned long long)seed;
    for (int t = 0; t < RT; ++t)
        for (int k = 0; k < KW; ++k) a[t][k] = RANDOM ? rnd_double(st) : seed * (threadIdx.x % 7 + t + k + 1) * 1e-3;
    for (int k = 0; k < KW; ++k) b[k] = RANDOM ? rnd_double(st) : seed * (threadIdx.x % 5 + k + 1) * 1e-3;
    double sink = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < batches; ++it) {
        d4 acc[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < KW; ++k)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][k], b[k], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) sink += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
        const double b0 = b[0];     // next batch: other entries of r
#pragma unroll
        for (int k = 0; k + 1 < KW; ++k) b[k] = b[k + 1];
        b[KW - 1] = b0 * 0.999;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (sink == 1.2345e300) out[0] = sink;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int RT, int KW, bool RANDOM>
static void run_tile(int wg_per_cu, int batches) {
    double* out;
    long long* cyc;
    hipMalloc(&out, 8);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_tile<RT, KW, RANDOM>), dim3(grid), dim3(256), 0, 0, out, cyc, batches, 1.0 + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long h = 0;
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double n_mfma_wave = (double)batches * RT * KW;
        const double flop = n_mfma_wave * 2048.0 * 4 * grid;
        if (rep == 2)
            printf("apply-shaped batches (%d x %d fragments of A), %s operands, workgroups per CU %d: %.1f TFLOP/s, %.1f counter ticks per MFMA and wave (%.3f ms)\n",
                   RT, KW, RANDOM ? "random" : "simple", wg_per_cu, flop / (ms * 1e9), (double)h / n_mfma_wave, ms);
    }
    hipFree(out);
    hipFree(cyc);
}

template <int NACC, bool RANDOM>
static void run(int wg_per_cu, int iters) {
    double* out;
    long long* cyc;
    hipMalloc(&out, 8);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_mfma<NACC, RANDOM>), dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0 + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long h = 0;
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double n_mfma_wave = (double)iters * 4 * NACC;
        const double flop = n_mfma_wave * 2048.0 * 4 * grid;
        if (rep == 2)
            printf("%s operands, accumulators %d, workgroups per CU %d: %.1f TFLOP/s, %.1f counter ticks per MFMA and wave (%.3f ms; counter %.0f MHz)\n",
                   RANDOM ? "random" : "simple", NACC, wg_per_cu, flop / (ms * 1e9), (double)h / n_mfma_wave, ms, (double)h / (ms * 1e3));
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    const int iters = 20000;
    run<1, false>(1, iters);
    run<4, false>(1, iters);
    run<4, false>(2, iters);
    run<8, false>(2, iters);
    run<4, true>(1, iters);
    run<4, true>(2, iters);
    run<8, true>(2, iters);
    run<4, true>(2, 10 * iters);
    run_tile<4, 12, false>(2, 20000);
    run_tile<4, 12, true>(2, 20000);
    run_tile<4, 12, true>(1, 20000);
    return 0;
}
