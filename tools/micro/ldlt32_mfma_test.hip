// Standalone check + timing of block_ldlt32_mfma (csrc/ldlt32.hpp) against a host LDL' in double:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/ldlt32_mfma_test tools/micro/ldlt32_mfma_test.hip && ./tools/micro/ldlt32_mfma_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

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
#include "../../multigridbarrier.jl_amd/csrc/ldlt32.hpp"

__global__ __launch_bounds__(256) void test_kernel(const double* A, int nb, double* Lout, double* dout, long long* cyc, int32_t* status, double* Wout) {
    __shared__ double Dn[32][33];
    __shared__ double Wv[32][33];
    __shared__ double Wv2[32][33];
    __shared__ double Tm[16][17];
    __shared__ double dq[32];
    const int tid = threadIdx.x;
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = tid; i < 32 * 32; i += 256) {
            const int rr = i % 32, c = i / 32;
            Dn[rr][c] = (rr >= c && rr < nb) ? A[rr * 32 + c] : 0.0;
        }
        __syncthreads();
        const long long t0 = clock64();
        block_ldlt32_inv_mfma(Dn, dq, nb, tid, Wv, status);
        const long long t1 = clock64();
        if (tid == 0) cyc[rep] = t1 - t0;
        __syncthreads();
        const long long t2 = clock64();
        block_inverse32_mfma(Dn, Wv2, Tm, tid);
        const long long t3 = clock64();
        if (tid == 0) cyc[3] = t3 - t2;
        __syncthreads();
    }
    {   // the factorization alone (what the solver calls), on a scratch copy
        __shared__ double D2[32][33];
        __shared__ double dq2[32];
        for (int i = tid; i < 32 * 32; i += 256) { const int rr = i % 32, c = i / 32; D2[rr][c] = (rr >= c && rr < nb) ? A[rr * 32 + c] : 0.0; }
        __syncthreads();
        const long long t4 = clock64();
        block_ldlt32_mfma(D2, dq2, nb, tid, nullptr);
        const long long t5 = clock64();
        if (tid == 0) cyc[2] = t5 - t4;
        __syncthreads();
    }
    for (int i = tid; i < 32 * 32; i += 256) { Lout[i] = Dn[i / 32][i % 32]; Wout[i] = Wv[i / 32][i % 32]; Wout[1024 + i] = Wv2[i / 32][i % 32]; }
    if (tid < 32) dout[tid] = dq[tid];
}

int main() {
    int fails = 0;
    for (int nb : {32, 31, 17, 16, 5, 1}) {
        std::vector<double> A(32 * 32, 0.0), L(32 * 32, 0.0), d(32, 1.0);
        // symmetric quasi-definite test matrix (LDL' without pivoting is stable): SPD block with a negative border pivot
        unsigned s = 12345u + nb;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / 16777216.0 - 0.5; };
        for (int r = 0; r < nb; ++r)
            for (int c = 0; c <= r; ++c) A[r * 32 + c] = A[c * 32 + r] = rnd();
        for (int r = 0; r < nb; ++r) A[r * 32 + r] += (r == nb - 1 && nb > 1) ? -40.0 : 20.0;
        // host reference
        std::vector<double> M(A);
        for (int j = 0; j < nb; ++j) {
            d[j] = M[j * 32 + j];
            for (int r = j + 1; r < nb; ++r) L[r * 32 + j] = M[r * 32 + j] / d[j];
            for (int r = j + 1; r < nb; ++r)
                for (int c = j + 1; c <= r; ++c) M[r * 32 + c] -= L[r * 32 + j] * d[j] * L[c * 32 + j];
        }
        double *dA, *dL, *dd, *dW; long long* dc; int32_t* ds;
        hipMalloc(&dA, 8 * 1024); hipMalloc(&dL, 8 * 1024); hipMalloc(&dd, 8 * 32); hipMalloc(&dc, 8 * 4); hipMalloc(&ds, 4); hipMalloc(&dW, 8 * 2048);
        hipMemcpy(dA, A.data(), 8 * 1024, hipMemcpyHostToDevice);
        hipMemset(ds, 0, 4);
        hipLaunchKernelGGL(test_kernel, dim3(1), dim3(256), 0, 0, dA, nb, dL, dd, dc, ds, dW);
        std::vector<double> gW(2048);
        hipMemcpy(gW.data(), dW, 8 * 2048, hipMemcpyDeviceToHost);
        std::vector<double> gL(1024), gd(32); long long cyc[4] = {0, 0, 0, 0}; int32_t st;
        hipMemcpy(gL.data(), dL, 8 * 1024, hipMemcpyDeviceToHost); hipMemcpy(gd.data(), dd, 8 * 32, hipMemcpyDeviceToHost);
        hipMemcpy(cyc, dc, 8 * 4, hipMemcpyDeviceToHost); hipMemcpy(&st, ds, 4, hipMemcpyDeviceToHost);
        double eL = 0.0, ed = 0.0, up = 0.0;
        for (int r = 0; r < 32; ++r) {
            for (int c = 0; c < 32; ++c) {
                const double ref = (c < r && r < nb) ? L[r * 32 + c] : 0.0;
                const double e = fabs(gL[r * 32 + c] - ref);
                if (c < r) eL = fmax(eL, e); else up = fmax(up, e);
            }
            ed = fmax(ed, fabs(gd[r] - (r < nb ? d[r] : 1.0)) / fabs(r < nb ? d[r] : 1.0));
        }
        double eW = 0.0;      // (I + L) W = I with the device's own L; W unit lower, zero above the diagonal
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double acc = gW[r * 32 + c];
                for (int k = 0; k < r; ++k) acc += gL[r * 32 + k] * gW[k * 32 + c];
                eW = fmax(eW, fabs(acc - (r == c ? 1.0 : 0.0)));
                if (c > r) eW = fmax(eW, fabs(gW[r * 32 + c]));
            }
        double eW2 = 0.0;     // the stand-alone matrix-core inverse of the same L
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double acc = gW[1024 + r * 32 + c];
                for (int k = 0; k < r; ++k) acc += gL[r * 32 + k] * gW[1024 + k * 32 + c];
                eW2 = fmax(eW2, fabs(acc - (r == c ? 1.0 : 0.0)));
                if (c > r) eW2 = fmax(eW2, fabs(gW[1024 + r * 32 + c]));
            }
        const bool ok = eL < 1e-13 && ed < 1e-13 && up == 0.0 && st == 0 && eW < 1e-13 && eW2 < 1e-13;
        printf("nb=%2d  max|L-Lref|=%.2e  max rel|d-dref|=%.2e  |LW-I|=%.2e  |LW2-I|=%.2e  upper/diag residue=%.1e  status=%d  cycles: with pipelined inverse %lld %lld, factorization alone %lld, inverse alone %lld  %s\n", nb, eL, ed, eW, eW2, up, st,
               cyc[0], cyc[1], cyc[2], cyc[3], ok ? "ok" : "FAIL");
        fails += !ok;
        hipFree(dA); hipFree(dL); hipFree(dd); hipFree(dc); hipFree(ds); hipFree(dW);
    }
    // a zero pivot must raise the status flag
    {
        std::vector<double> A(32 * 32, 0.0);
        for (int r = 0; r < 32; ++r) A[r * 32 + r] = (r == 7) ? 0.0 : 3.0;
        double *dA, *dL, *dd, *dW; long long* dc; int32_t* ds;
        hipMalloc(&dA, 8 * 1024); hipMalloc(&dL, 8 * 1024); hipMalloc(&dd, 8 * 32); hipMalloc(&dc, 8 * 4); hipMalloc(&ds, 4); hipMalloc(&dW, 8 * 2048);
        hipMemcpy(dA, A.data(), 8 * 1024, hipMemcpyHostToDevice); hipMemset(ds, 0, 4);
        hipLaunchKernelGGL(test_kernel, dim3(1), dim3(256), 0, 0, dA, 32, dL, dd, dc, ds, dW);
        int32_t st; hipMemcpy(&st, ds, 4, hipMemcpyDeviceToHost);
        printf("zero pivot: status=%d %s\n", st, st == 1 ? "ok" : "FAIL");
        fails += st != 1;
    }
    return fails;
}
