// Calibration: sustained v_mfma_f64_16x16x4_f64 rate on this GPU (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    double4_t acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
    }
    double s = 0;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 4; ++r) s += acc[u][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    double* d; hipMalloc(&d, 8 * 256 * 4096);
    for (int wgs : {256, 512, 1024, 2048}) {
        const int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<wgs, 256>>>(d, 100, 1.0, 1.0); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<wgs, 256>>>(d, iters, 1.0, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)wgs * 4 /*waves*/ * iters * 4 /*mfma*/ * 2048.0;
        printf("wgs %d: %.2f ms  %.1f TFLOP/s\n", wgs, ms, flop / ms / 1e9);
    }
    return 0;
}
