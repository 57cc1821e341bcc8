// Latency microbenchmarks for the in-wave LDL' pivot chain on gfx950 (one wave, shader cycles):
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/chain_probe.hip -o tools/micro/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), srclane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double fast_recip(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, x, 1.0);
    x = __builtin_fma(x, e, x);
    e = __builtin_fma(-d, x, 1.0);
    x = __builtin_fma(x, e, x);
    return x;
}

__global__ void probe(double* out, long long* cyc, double seed) {
    __shared__ double buf[64];
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3;
    long long t0, t1;
#define TICK(t) do { asm volatile("s_nop 0" :: "v"(x) : "memory"); t = clock64(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
    // (a) 64 dependent fast_recip
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = fast_recip(x) + 1.0;
    TICK(t1);
    if (lane == 0) cyc[0] = t1 - t0;
    // (b) 64 dependent FMAs
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = __builtin_fma(x, 1.0000001, 0.5);
    TICK(t1);
    if (lane == 0) cyc[1] = t1 - t0;
    // (c) 64 dependent readlane_f64 + FMA
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = __builtin_fma(x, 0.999, readlane_f64(x, i & 31));
    TICK(t1);
    if (lane == 0) cyc[2] = t1 - t0;
    // (d) 64 dependent v_rcp_f64 only
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_rcp(x) + 1.5;
    TICK(t1);
    if (lane == 0) cyc[3] = t1 - t0;
    // (e) LDS write -> read round trip, dependent, 64 times
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        buf[lane] = x;
        x = buf[(lane + 1) & 63] * 0.5 + 1.0;
    }
    TICK(t1);
    if (lane == 0) cyc[4] = t1 - t0;
    // (f) 512 independent FMAs (16 accumulators)
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = x + k;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_fma(acc[k], 1.0000001, x);
    asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[5]), "v"(acc[10]), "v"(acc[15]) : "memory");
    TICK(t1);
    if (lane == 0) cyc[5] = t1 - t0;
#pragma unroll
    for (int k = 0; k < 16; ++k) x += acc[k];
    // (g) isfinite / compare chain
    int badc = 0;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 64; ++i) { x = x * 1.0000001; if (x == 0.0 || !isfinite(x)) badc++; }
    TICK(t1);
    if (lane == 0) cyc[6] = t1 - t0;
    out[lane] = x + badc;
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8 * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
        hipDeviceSynchronize();
    }
    long long h[8];
    hipMemcpy(h, cyc, 8 * 8, hipMemcpyDeviceToHost);
    const char* nm[] = {"fast_recip+add chain x64", "dependent fma x64", "readlane_f64+fma x64", "rcp+add x64", "LDS write->read x64",
                        "512 independent fma", "mul+cmp/isfinite x64"};
    for (int i = 0; i < 7; ++i) printf("%-28s %8lld cycles  (%.1f per item)\n", nm[i], h[i], h[i] / (i == 5 ? 512.0 : 64.0));
    return 0;
}
