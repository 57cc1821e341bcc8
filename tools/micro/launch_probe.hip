// Launch-overhead microbenchmark: hipcc -O3 --offload-arch=gfx950 tools/micro/launch_probe.hip -o tools/micro/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_small(double* p) { extern __shared__ double sh[]; if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] = sh[0]; }
__global__ __launch_bounds__(1024) void k_big(double* p) { extern __shared__ double sh[]; if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] = sh[0]; }
__global__ __launch_bounds__(1024) void k_big_attr(double* p) { extern __shared__ double sh[]; if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] = sh[0]; }
template <class F> float timeit(F f, int n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return 1e3f * ms / n;
}
int main() {
    hipFuncSetAttribute((const void*)k_big_attr, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024);
    for (int grid : {1, 16, 196}) {
        float a = timeit([&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, 0, (double*)nullptr); }, 200);
        float b = timeit([&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 18 * 1024, 0, (double*)nullptr); }, 200);
        float c = timeit([&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(1024), 0, 0, (double*)nullptr); }, 200);
        float d = timeit([&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(1024), 18 * 1024, 0, (double*)nullptr); }, 200);
        float e = timeit([&] { hipLaunchKernelGGL(k_big_attr, dim3(grid), dim3(1024), 18 * 1024, 0, (double*)nullptr); }, 200);
        float f = timeit([&] { hipLaunchKernelGGL(k_big_attr, dim3(grid), dim3(1024), 100 * 1024, 0, (double*)nullptr); }, 200);
        printf("grid %3d: 256thr/0KB %.1f us | 256thr/18KB %.1f | 1024thr/0KB %.1f | 1024thr/18KB %.1f | 1024thr/18KB(attr 130KB) %.1f | 1024thr/100KB(attr) %.1f\n", grid, a, b, c, d, e, f);
    }
    return 0;
}
