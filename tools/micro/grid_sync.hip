// Calibration: cost of a cooperative-groups grid barrier vs a kernel boundary on this GPU.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ __launch_bounds__(256) void ksync(int iters, double* out) {
    cg::grid_group g = cg::this_grid();
    double s = 0;
    for (int i = 0; i < iters; ++i) { s += i; g.sync(); }
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void kempty(double* out) { if (threadIdx.x == 0) out[blockIdx.x] = 1.0; }
int main() {
    double* d; hipMalloc(&d, 8 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {4, 32, 128, 256, 512}) {
        int iters = 2000;
        void* args[] = {&iters, &d};
        hipLaunchCooperativeKernel((void*)ksync, dim3(wgs), dim3(256), args, 0, 0); hipDeviceSynchronize();
        hipEventRecord(e0);
        hipError_t err = hipLaunchCooperativeKernel((void*)ksync, dim3(wgs), dim3(256), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid.sync  wgs %4d: %.2f us per sync (%s)\n", wgs, 1e3 * ms / iters, hipGetErrorString(err));
        hipEventRecord(e0);
        for (int i = 0; i < 2000; ++i) kempty<<<wgs, 256>>>(d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("kernel boundary wgs %4d: %.2f us per launch\n", wgs, 1e3 * ms / 2000);
    }
    return 0;
}
