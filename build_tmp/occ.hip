#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double* p) { extern __shared__ double s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
    for (int bytes : {20480, 21760, 23040, 23552, 24320, 25600, 25920, 26880, 26944, 27136, 28160, 32768}) {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, (size_t)bytes);
        printf("dynamic LDS %d B -> %d workgroups per CU\n", bytes, nb);
    }
    return 0;
}
