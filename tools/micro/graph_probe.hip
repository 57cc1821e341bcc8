// Launch overhead of a chain of small dependent kernels: individual launches vs one hipGraph launch (captured once).
// Build: hipcc --offload-arch=gfx950 -O3 -o graph_probe graph_probe.hip ; run: ./graph_probe [kernels] [work]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void small(double* p, int work) {
    double v = p[threadIdx.x];
    for (int i = 0; i < work; ++i) v = v * 1.0000001 + 1e-9;
    p[threadIdx.x] = v;
}
int main(int argc, char** argv) {
    const int nk = argc > 1 ? atoi(argv[1]) : 80, work = argc > 2 ? atoi(argv[2]) : 200, reps = 200;
    double* d; CK(hipMalloc(&d, 256 * sizeof(double))); CK(hipMemset(d, 0, 256 * sizeof(double)));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, st, d, work);
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, st, d, work);
        CK(hipStreamSynchronize(st));
    }
    const double t_stream = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, st, d, work);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
    const double t_graph = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%d kernels, work %d: stream %.1f us (%.2f us / kernel), graph %.1f us (%.2f us / kernel)\n", nk, work, 1e6 * t_stream,
           1e6 * t_stream / nk, 1e6 * t_graph, 1e6 * t_graph / nk);
    return 0;
}
