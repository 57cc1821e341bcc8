// Accuracy check of the register-resident LDL' (wave_ldlt_regs, as used by mf_factor_wave) on graded SPD matrices:
//   hipcc --offload-arch=gfx950 -O3 -o wave_ldlt_test wave_ldlt_test.hip && ./wave_ldlt_test
// The bordered matrix [A -b; -b' -1] is factored by one wave (lane r = row r), the host runs the backward sweep
// x = L'^{-1} e_n and compares lambda^2 = b'x with a long-double elimination.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
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
template <int NBT>
__device__ __forceinline__ bool wave_ldlt_regs(double (&a)[NBT], int nb, int lane) {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
        if (j < nb) {
            const double d = readlane_f64(a[j], j);
            if (d == 0.0 || !isfinite(d)) bad = true;
            const double inv = fast_recip(d);
            const double aj = a[j];
            const double lr = aj * inv;
#pragma unroll
            for (int c = j + 1; c < NBT; ++c) {
                const double v = readlane_f64(aj, c);
                a[c] -= lr * v;
            }
            if (lane > j) a[j] = lr;
        }
    }
    return bad;
}
constexpr int MW = 48;
__global__ void factor(const double* in, double* out, int m, int k) {
    const int lane = threadIdx.x;
    double a[MW];
#pragma unroll
    for (int c = 0; c < MW; ++c) a[c] = (lane < m && c <= lane) ? in[lane + c * m] : 0.0;
    wave_ldlt_regs<MW>(a, k, lane);
    if (lane < m)
#pragma unroll
        for (int c = 0; c < MW; ++c)
            if (c <= lane) out[lane + c * m] = a[c];
}

int main() {
    const int n = 37, m = n + 1;
    std::mt19937_64 gen(1);
    std::normal_distribution<double> N01;
    for (double M : {1e6, 1e10, 1e13}) {
        double worst = 0, worst_host = 0;
        for (int trial = 0; trial < 10; ++trial) {
            std::vector<double> B(n * n), A(n * n, 0.0), V(2 * n), b(n);
            for (auto& x : B) x = N01(gen);
            for (auto& x : V) x = N01(gen);
            for (auto& x : b) x = 1e5 * N01(gen);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double s = 0;
                    for (int q = 0; q < n; ++q) s += B[i * n + q] * B[j * n + q];
                    A[i + j * n] = s / n + (i == j ? 0.1 : 0.0) + M * (V[i] * V[j] + V[n + i] * V[n + j]);
                }
            // bordered, lower triangle, column-major m x m
            std::vector<double> W(m * m, 0.0);
            for (int j = 0; j < n; ++j)
                for (int i = j; i < n; ++i) W[i + j * m] = A[i + j * n];
            for (int j = 0; j < n; ++j) W[n + j * m] = -b[j];
            W[n + n * m] = -1.0;
            double *din, *dout;
            hipMalloc(&din, m * m * 8); hipMalloc(&dout, m * m * 8);
            hipMemcpy(din, W.data(), m * m * 8, hipMemcpyHostToDevice);
            hipMemset(dout, 0, m * m * 8);
            hipLaunchKernelGGL(factor, dim3(1), dim3(64), 0, 0, din, dout, m, n);
            std::vector<double> F(m * m);
            hipMemcpy(F.data(), dout, m * m * 8, hipMemcpyDeviceToHost);
            hipFree(din); hipFree(dout);
            // device factors: backward sweep on the host
            auto lam = [&](const std::vector<double>& Fm) {
                std::vector<double> x(m, 0.0); x[n] = 1.0;
                for (int j = n - 1; j >= 0; --j) { double s = 0; for (int r = j + 1; r < m; ++r) s += Fm[r + j * m] * x[r]; x[j] = -s; }
                double l = 0; for (int i = 0; i < n; ++i) l += b[i] * x[i];
                return l;
            };
            // the same textbook elimination on the host (double)
            std::vector<double> H(W);
            for (int j = 0; j < n; ++j) {
                const double d = H[j + j * m];
                for (int c = j + 1; c < m; ++c) { const double lr_c = H[c + j * m]; for (int r = c; r < m; ++r) H[r + c * m] -= (H[r + j * m] / d) * lr_c; }
                for (int r = j + 1; r < m; ++r) H[r + j * m] /= d;
            }
            // long double reference
            std::vector<long double> Mx((size_t)n * (n + 1));
            for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) Mx[i * (n + 1) + j] = A[i + j * n]; Mx[i * (n + 1) + n] = b[i]; }
            for (int j = 0; j < n; ++j) for (int i = j + 1; i < n; ++i) { long double f = Mx[i * (n + 1) + j] / Mx[j * (n + 1) + j]; for (int q = j; q <= n; ++q) Mx[i * (n + 1) + q] -= f * Mx[j * (n + 1) + q]; }
            std::vector<long double> xl(n);
            for (int j = n - 1; j >= 0; --j) { long double s = Mx[j * (n + 1) + n]; for (int q = j + 1; q < n; ++q) s -= Mx[j * (n + 1) + q] * xl[q]; xl[j] = s / Mx[j * (n + 1) + j]; }
            long double lref = 0; for (int i = 0; i < n; ++i) lref += (long double)b[i] * xl[i];
            worst = std::fmax(worst, std::fabs((lam(F) - (double)lref) / (double)lref));
            worst_host = std::fmax(worst_host, std::fabs((lam(H) - (double)lref) / (double)lref));
        }
        printf("M=%g: worst relative error of lambda^2: device wave_ldlt_regs<48> %.2e, host textbook %.2e\n", M, worst, worst_host);
    }
    return 0;
}
