// dense.hip -- dense-operator (spectral) path: GEMVs, node-parallel cone kernel, and the
// symmetric fp64 MFMA GEMM that forms H = (D R)' Ybar (D R).  See dense.hpp.
#include <hip/hip_runtime.h>

#include "dense.hpp"

namespace mgbhip {

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int dtri_index(int k, int k2, int NY) {   // k <= k2
    return k * NY - (k * (k - 1)) / 2 + (k2 - k);
}

// ---------------------------------------------------------------------------------------------
// GEMV.  y = A x: a workgroup of 16 waves owns 64 rows; wave w sums its column range with the
// lane on the row (coalesced column-major reads), the 16 partials are added in fixed order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dense_gemv_n_kernel(int rows, int cols, const double* __restrict__ A,
                                                            int64_t lda, const double* __restrict__ x,
                                                            double* __restrict__ y) {
    __shared__ double part[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + lane;
    const int cw = (cols + 15) / 16;
    const int c0 = w * cw;
    int c1 = c0 + cw;
    if (c1 > cols) c1 = cols;
    double s = 0.0;
    if (row < rows) {
        const double* Ar = A + row;
        int c = c0;
        for (; c + 4 <= c1; c += 4) {
            const double a0 = Ar[lda * c], a1 = Ar[lda * (c + 1)], a2 = Ar[lda * (c + 2)], a3 = Ar[lda * (c + 3)];
            s += a0 * x[c];
            s += a1 * x[c + 1];
            s += a2 * x[c + 2];
            s += a3 * x[c + 3];
        }
        for (; c < c1; ++c) s += Ar[lda * c] * x[c];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && row < rows) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][lane];
        y[row] = t;
    }
}

// y (+)= A' x: one wave per column, lanes stride the rows, fixed shuffle tree.
template <bool ADD>
__global__ __launch_bounds__(256) void dense_gemv_t_kernel(int rows, int cols, const double* __restrict__ A,
                                                           int64_t lda, const double* __restrict__ x,
                                                           double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= cols) return;
    const double* Ac = A + lda * col;
    double s = 0.0;
    for (int r = lane; r < rows; r += 64) s += Ac[r] * x[r];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[col] = ADD ? y[col] + s : s;
}

__global__ __launch_bounds__(256) void dense_transpose_kernel(int n, const double* __restrict__ A,
                                                              double* __restrict__ At) {
    __shared__ double tile[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + tx, j = blockIdx.y * 16 + ty;
    if (i < n && j < n) tile[ty][tx] = A[i + (int64_t)n * j];
    __syncthreads();
    const int i2 = blockIdx.y * 16 + tx, j2 = blockIdx.x * 16 + ty;     // At[i2, j2] = A[j2, i2]
    if (i2 < n && j2 < n) At[i2 + (int64_t)n * j2] = tile[tx][ty];
}

// ---------------------------------------------------------------------------------------------
// node kernel: one thread per node, Dz rows already formed by the GEMVs
// ---------------------------------------------------------------------------------------------
template <int NY, int MODE>
__global__ __launch_bounds__(256) void dense_node_kernel(const ElemParams P) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int64_t n = P.n;
    const int64_t node = (int64_t)blockIdx.x * 256 + tid;
    const bool active = node < n;
    double y[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) y[k] = active ? P.dn_Dz[node + n * k] : 0.0;
    double F = 0.0;
    double g[NY];
    double H[NY * NY];
    (void)g;
    (void)H;
    if (MODE == MODE_NODE_F) {
        if (active) {
            cone_eval<NY, 0>(P.cone, node, n, y, F, g, H);
            P.out_F[node] = F;
            if (P.out_Dz != nullptr) {
#pragma unroll
                for (int k = 0; k < NY; ++k) P.out_Dz[node + n * k] = y[k];
            }
        }
        return;
    }
    if (MODE == MODE_NODE_SLACK) {
        if (active) P.out_F[node] = cone_slack<NY>(P.cone, node, n, y);
        return;
    }
    if (MODE == MODE_F0) {
        double val = 0.0;
        if (active) {
            cone_eval<NY, 0>(P.cone, node, n, y, F, g, H);
            double bar;
            if (P.bw != nullptr) {
                const double bwv = P.bw[node];
                bar = (bwv == 0.0) ? 0.0 : bwv * F;
            } else {
                bar = (P.invn == 0.0) ? 0.0 : P.invn * F;   // invn == 0: linear part only (c_dot_Dz)
            }
            double lin = 0.0;
#pragma unroll
            for (int k = 0; k < NY; ++k) lin += P.c[node + n * k] * y[k];
            val = bar + P.w[node] * lin;
        }
        red[tid] = val;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) red[tid] += red[tid + off];
            __syncthreads();
        }
        if (tid == 0) P.out_partial[blockIdx.x] = red[0];
        return;
    }
    if (MODE == MODE_F1) {
        if (active) {
            cone_eval<NY, 1>(P.cone, node, n, y, F, g, H);
            const double wv = P.w[node];
            const double bwv = P.bw ? P.bw[node] : 0.0;
#pragma unroll
            for (int k = 0; k < NY; ++k) {
                const double sc = P.bw ? ((bwv == 0.0) ? 0.0 : bwv * g[k]) : P.invn * g[k];
                P.dn_Y[node + n * k] = sc + wv * P.c[node + n * k];
            }
        }
        return;
    }
    if (MODE == MODE_F2) {
        if (active) {
            cone_eval<NY, 2>(P.cone, node, n, y, F, g, H);
            const double bwv = P.bw ? P.bw[node] : 0.0;
#pragma unroll
            for (int k = 0; k < NY; ++k)
#pragma unroll
                for (int k2 = k; k2 < NY; ++k2) {
                    const double h = H[k * NY + k2];
                    const double sc = P.bw ? ((bwv == 0.0) ? 0.0 : bwv * h) : P.invn * h;
                    P.dn_Y[node + n * dtri_index(k, k2, NY)] = sc;
                }
        }
        return;
    }
}

template <int NY>
void launch_node_ny(const ElemParams& P, int mode, hipStream_t st) {
    const dim3 grid((unsigned)dense_grid(P.n)), blk(256);
    switch (mode) {
        case MODE_F0: hipLaunchKernelGGL((dense_node_kernel<NY, MODE_F0>), grid, blk, 0, st, P); break;
        case MODE_F1: hipLaunchKernelGGL((dense_node_kernel<NY, MODE_F1>), grid, blk, 0, st, P); break;
        case MODE_F2: hipLaunchKernelGGL((dense_node_kernel<NY, MODE_F2>), grid, blk, 0, st, P); break;
        case MODE_NODE_F: hipLaunchKernelGGL((dense_node_kernel<NY, MODE_NODE_F>), grid, blk, 0, st, P); break;
        case MODE_NODE_SLACK: hipLaunchKernelGGL((dense_node_kernel<NY, MODE_NODE_SLACK>), grid, blk, 0, st, P); break;
        default: throw InvalidArgument("launch_dense_eval: bad mode");
    }
    MGB_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// W = Ybar * DR: thread per (node, column)
// ---------------------------------------------------------------------------------------------
template <int NY>
__global__ __launch_bounds__(256) void dense_weight_kernel(int klo, int khi, int64_t n, int64_t ld,
                                                           const double* __restrict__ DR,
                                                           const double* __restrict__ Yh, double* __restrict__ W) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t j = blockIdx.y;
    double dr[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) dr[k] = (k >= klo && k <= khi) ? DR[k * n + i + ld * j] : 0.0;
#pragma unroll
    for (int k = 0; k < NY; ++k) {
        if (k < klo || k > khi) continue;
        double acc = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < NY; ++k2) {
            if (k2 < klo || k2 > khi) continue;
            const int t = (k <= k2) ? dtri_index(k, k2, NY) : dtri_index(k2, k, NY);
            acc += Yh[i + n * t] * dr[k2];
        }
        W[k * n + i + ld * j] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// C (+)= A' diag(v) B on the fp64 matrix cores.  64 x 64 tile per workgroup, 4 waves as 2 x 2,
// each wave 2 x 2 MFMA tiles of v_mfma_f64_16x16x4_f64; K tiles of 16 double-buffered in LDS
// with the next tile's global loads in flight during the MFMAs.  Both operands are contiguous
// along K in memory.  The MFMA is issued with (B-fragment, A-fragment) so that the lane index
// of a result is the ROW of C and the stores are coalesced along columns of the column-major C.
// Fragment maps (cdna guide, f64 16x16x4): a: [lane&15][k = lane>>4], b: [k = lane>>4][lane&15],
// d: col = lane&15, row = (lane>>4) + 4*reg.
// ---------------------------------------------------------------------------------------------
constexpr int GT = 64;     // tile edge
constexpr int GK = 16;     // K tile

template <bool HASV>
__global__ __launch_bounds__(256) void dense_gemm_tn_kernel(int M, int N, int K, const double* __restrict__ A,
                                                            int64_t lda, const double* __restrict__ v,
                                                            const double* __restrict__ B, int64_t ldb,
                                                            double* __restrict__ C, int64_t ldc, int accumulate,
                                                            int symmetric) {
    __shared__ double As[2][GK][GT + 1];      // odd row stride: the k-fastest stores below are conflict-free
    __shared__ double Bs[2][GK][GT + 1];
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (symmetric && bi > bj) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int i0 = bi * GT, j0 = bj * GT;
    // Staging map: 16 consecutive lanes read 16 consecutive k of one column (one full 128-byte
    // line), each thread covers 4 columns 16 apart; a wave-level load touches 4 lines, not 16.
    // Branch-free: addresses are clamped into range so every load is unconditional and nothing
    // consumes a loaded value before the LDS store (a guarded load or an early use makes the
    // compiler drain vmcnt inside the prefetch); masks and the diag(v) scaling apply at store time.
    static_assert(GK == 16 && GT == 64, "staging map assumes a 16 x 64 operand tile");
    const int lk = tid & 15, lc = tid >> 4;
    const double* Ap[4];
    const double* Bp[4];
    bool a_in[4], b_in[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a_in[q] = i0 + lc + 16 * q < M;
        b_in[q] = j0 + lc + 16 * q < N;
        Ap[q] = A + lda * (int64_t)min(i0 + lc + 16 * q, M - 1);
        Bp[q] = B + ldb * (int64_t)min(j0 + lc + 16 * q, N - 1);
    }
    // Register ring of three K tiles in flight ahead of the one being multiplied (one tile of
    // look-ahead is 0.4 us of MFMA work, an L2/MALL round trip is several times that).
    double ra[3][4], rb[3][4], rv[3] = {1.0, 1.0, 1.0};
    auto gload = [&](int k0, double (&xa)[4], double (&xb)[4], double& xv) {
        const int kc = min(k0 + lk, K - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xa[q] = Ap[q][kc];
            xb[q] = Bp[q][kc];
        }
        if (HASV) xv = v[kc];
    };
    auto sstore = [&](int buf, int k0, const double (&xa)[4], const double (&xb)[4], double xv) {
        const bool kin = k0 + lk < K;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            As[buf][lk][lc + 16 * q] = (kin && a_in[q]) ? xa[q] : 0.0;
            Bs[buf][lk][lc + 16 * q] = (kin && b_in[q]) ? (HASV ? xb[q] * xv : xb[q]) : 0.0;
        }
    };
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int nkt = (K + GK - 1) / GK;
    const int fr = lane & 15, fk = lane >> 4;
    auto multiply = [&](int cur) {
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            const double a0 = As[cur][ks * 4 + fk][wm * 32 + fr];
            const double a1 = As[cur][ks * 4 + fk][wm * 32 + 16 + fr];
            const double b0 = Bs[cur][ks * 4 + fk][wn * 32 + fr];
            const double b1 = Bs[cur][ks * 4 + fk][wn * 32 + 16 + fr];
            // operands swapped: result(lane&15 -> row of C within the a-tile, reg/lane>>4 -> column)
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
        }
    };
    gload(0, ra[0], rb[0], rv[0]);
    gload(GK, ra[1], rb[1], rv[1]);
    gload(2 * GK, ra[2], rb[2], rv[2]);
    sstore(0, 0, ra[0], rb[0], rv[0]);
    __syncthreads();
    // tile t is in LDS buffer t & 1; tiles t+1, t+2 wait in ring slots (t+1)%3, (t+2)%3; slot t%3
    // is free and receives tile t+3.  Unrolled by 3: the ring slots are compile-time registers.
    for (int kt = 0; kt < nkt; kt += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = kt + u;
            if (t < nkt) {
                gload((t + 3) * GK, ra[u], rb[u], rv[u]);
                multiply(t & 1);
                if (t + 1 < nkt) sstore((t + 1) & 1, (t + 1) * GK, ra[(u + 1) % 3], rb[(u + 1) % 3], rv[(u + 1) % 3]);
                __syncthreads();
            }
        }
    }
    // D'[jj][ii] with jj = (lane>>4) + 4*reg (row of the swapped product = column of C),
    // ii = lane&15 (column of the swapped product = row of C)
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wm * 32 + ta * 16 + fr;
                const int col = j0 + wn * 32 + tb * 16 + fk + 4 * r;
                if (row < M && col < N) {
                    double val = acc[ta][tb][r];
                    if (accumulate) val += C[row + ldc * col];
                    C[row + ldc * col] = val;
                    if (symmetric && bi != bj) C[col + ldc * row] = val;
                }
            }
}

// ---- pivoted fallback of the symmetric solve -----------------------------------------------------------------------------
// `solve(symmetric(H), g)` in the reference is Julia's `Symmetric(H) \ g`: Cholesky, then LDL', then LU when the
// symmetric factorizations fail (src/utils.jl:142-145).  The multifrontal LDL' of mf_numeric.hip is un-pivoted (a fixed
// elimination order is what keeps its symbolic plan); when it meets an exactly zero / non-finite pivot on a system
// small enough to be held densely, this kernel plays the reference's last resort: dense LU with partial (row) pivoting
// of the matrix whose upper triangle is H's, one workgroup, the matrix in a row-major global scratch.  It is a rare
// path (a singular leading block of an otherwise regular coarse system), so it is written for determinism, not speed:
// pivot = the first row of largest modulus (ties to the smaller index), every sum in a fixed order.
constexpr int LU_THREADS = 1024;
__global__ __launch_bounds__(LU_THREADS) void dense_lu_solve_kernel(int m, const int32_t* __restrict__ Hptr,
                                                                    const int32_t* __restrict__ Hcol,
                                                                    const double* __restrict__ Hval, double* __restrict__ A,
                                                                    const double* __restrict__ g, double* __restrict__ x,
                                                                    int32_t* __restrict__ status) {
    __shared__ double rv[LU_THREADS];
    __shared__ int ri[LU_THREADS];
    __shared__ int piv_row;
    __shared__ double piv_val;
    const int tid = threadIdx.x;
    const int64_t mm = (int64_t)m * m;
    for (int64_t i = tid; i < mm; i += LU_THREADS) A[i] = 0.0;
    for (int i = tid; i < m; i += LU_THREADS) x[i] = g[i];
    __syncthreads();
    // symmetric(H): the upper triangle of the assembled CSR, mirrored
    for (int i = tid; i < m; i += LU_THREADS)
        for (int32_t q = Hptr[i]; q < Hptr[i + 1]; ++q) {
            const int j = Hcol[q];
            if (j >= i) {
                A[(int64_t)i * m + j] = Hval[q];
                A[(int64_t)j * m + i] = Hval[q];      // distinct (i, j) pairs: no two threads write one entry
            }
        }
    __syncthreads();
    for (int k = 0; k < m; ++k) {
        // pivot search in column k, rows k .. m-1
        double best = -1.0;
        int bi = m;
        for (int i = k + tid; i < m; i += LU_THREADS) {
            double v = fabs(A[(int64_t)i * m + k]);
            if (!(v == v)) v = INFINITY;                 // a NaN entry is reported through the pivot test below
            if (v > best) { best = v; bi = i; }
        }
        rv[tid] = best;
        ri[tid] = bi;
        __syncthreads();
        for (int off = LU_THREADS / 2; off > 0; off >>= 1) {
            if (tid < off) {
                const double a = rv[tid], b = rv[tid + off];
                if (b > a || (b == a && ri[tid + off] < ri[tid])) { rv[tid] = b; ri[tid] = ri[tid + off]; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            piv_row = ri[0];
            piv_val = (ri[0] < m) ? A[(int64_t)ri[0] * m + k] : 0.0;
            if (!(rv[0] > 0.0) || !isfinite(rv[0])) atomicOr(status, 1);      // singular to working precision / non-finite
        }
        __syncthreads();
        const int p = piv_row;
        const double d = piv_val;
        if (!(fabs(d) > 0.0) || !isfinite(d)) return;                          // uniform: every thread sees the same pivot
        if (p != k) {
            for (int j = tid; j < m; j += LU_THREADS) {
                const double a = A[(int64_t)k * m + j];
                A[(int64_t)k * m + j] = A[(int64_t)p * m + j];
                A[(int64_t)p * m + j] = a;
            }
            if (tid == 0) { const double a = x[k]; x[k] = x[p]; x[p] = a; }
            __syncthreads();
        }
        // multipliers into column k, right-hand side
        const double xk = x[k];
        __syncthreads();
        for (int i = k + 1 + tid; i < m; i += LU_THREADS) {
            const double l = A[(int64_t)i * m + k] / d;
            A[(int64_t)i * m + k] = l;
            x[i] -= l * xk;
        }
        __syncthreads();
        // rank-1 update of the trailing block (row-major: consecutive threads on consecutive columns)
        const int w = m - k - 1;
        const int64_t tot = (int64_t)w * w;
        for (int64_t t = tid; t < tot; t += LU_THREADS) {
            const int i = k + 1 + (int)(t / w), j = k + 1 + (int)(t % w);
            A[(int64_t)i * m + j] -= A[(int64_t)i * m + k] * A[(int64_t)k * m + j];
        }
        __syncthreads();
    }
    // back substitution U x = y
    for (int k = m - 1; k >= 0; --k) {
        double s = 0.0;
        for (int j = k + 1 + tid; j < m; j += LU_THREADS) s += A[(int64_t)k * m + j] * x[j];
        rv[tid] = s;
        __syncthreads();
        for (int off = LU_THREADS / 2; off > 0; off >>= 1) {
            if (tid < off) rv[tid] += rv[tid + off];
            __syncthreads();
        }
        if (tid == 0) x[k] = (x[k] - rv[0]) / A[(int64_t)k * m + k];
        __syncthreads();
    }
}

}  // namespace

int64_t dense_grid(int64_t n) { return (n + 255) / 256; }

void launch_dense_gemv_n(int rows, int cols, const double* A, int64_t lda, const double* x, double* y,
                         hipStream_t st) {
    if (rows == 0) return;
    hipLaunchKernelGGL(dense_gemv_n_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(1024), 0, st, rows, cols, A, lda,
                       x, y);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_dense_gemv_t(int rows, int cols, const double* A, int64_t lda, const double* x, double* y, bool add,
                         hipStream_t st) {
    if (cols == 0) return;
    const dim3 grid((unsigned)((cols + 3) / 4));
    if (add) hipLaunchKernelGGL(dense_gemv_t_kernel<true>, grid, dim3(256), 0, st, rows, cols, A, lda, x, y);
    else hipLaunchKernelGGL(dense_gemv_t_kernel<false>, grid, dim3(256), 0, st, rows, cols, A, lda, x, y);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_dense_transpose(int n, const double* A, double* At, hipStream_t st) {
    if (n == 0) return;
    const unsigned g = (unsigned)((n + 15) / 16);
    hipLaunchKernelGGL(dense_transpose_kernel, dim3(g, g), dim3(256), 0, st, n, A, At);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_dense_gemm_tn(int M, int N, int K, const double* A, int64_t lda, const double* v, const double* B,
                          int64_t ldb, double* C, int64_t ldc, bool accumulate, bool symmetric, hipStream_t st) {
    if (M == 0 || N == 0) return;
    if (symmetric && M != N) throw InvalidArgument("dense_gemm_tn: symmetric product must be square");
    const dim3 grid((unsigned)((M + GT - 1) / GT), (unsigned)((N + GT - 1) / GT));
    if (K == 0) throw InvalidArgument("dense_gemm_tn: empty contraction");
    if (v != nullptr)
        hipLaunchKernelGGL(dense_gemm_tn_kernel<true>, grid, dim3(256), 0, st, M, N, K, A, lda, v, B, ldb, C, ldc,
                           accumulate ? 1 : 0, symmetric ? 1 : 0);
    else
        hipLaunchKernelGGL(dense_gemm_tn_kernel<false>, grid, dim3(256), 0, st, M, N, K, A, lda, v, B, ldb, C, ldc,
                           accumulate ? 1 : 0, symmetric ? 1 : 0);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_dense_weight(int NY, int klo, int khi, int64_t n, int64_t m, int64_t ld, const double* DR,
                         const double* Yh, double* W, hipStream_t st) {
    if (n == 0 || m == 0) return;
    const dim3 grid((unsigned)dense_grid(n), (unsigned)m), blk(256);
    switch (NY) {
#define MGB_CASE(X) \
    case X: hipLaunchKernelGGL((dense_weight_kernel<X>), grid, blk, 0, st, klo, khi, n, ld, DR, Yh, W); break;
        MGB_CASE(1) MGB_CASE(2) MGB_CASE(3) MGB_CASE(4) MGB_CASE(5) MGB_CASE(6) MGB_CASE(7) MGB_CASE(8) MGB_CASE(9) MGB_CASE(10)
#undef MGB_CASE
        default: throw InvalidArgument("dense_weight: nD out of range");
    }
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_dense_eval(const ElemParams& P, int mode, hipStream_t st) {
    if (P.N != 1) throw InvalidArgument("dense path: one notional element expected");
    if (P.dn_Dz == nullptr || P.dn_Y == nullptr) throw InvalidArgument("dense path: workspace missing");
    const int n = (int)P.n;
    // Dz_k = D_k z_{state(k)}   (src/convex.jl:125 with dense D)
    for (int k = 0; k < P.nD; ++k) {
        const double* za = P.z0 + (int64_t)P.D_state[k] * n;
        const double* op = P.ops[P.D_op[k]];
        if (op == nullptr) MGB_HIP_CHECK(hipMemcpyAsync(P.dn_Dz + (int64_t)k * n, za, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
        else launch_dense_gemv_n(n, n, op, n, za, P.dn_Dz + (int64_t)k * n, st);
    }
    switch (P.nD) {
#define MGB_CASE(X) case X: launch_node_ny<X>(P, mode, st); break;
        MGB_CASE(1) MGB_CASE(2) MGB_CASE(3) MGB_CASE(4) MGB_CASE(5) MGB_CASE(6) MGB_CASE(7) MGB_CASE(8) MGB_CASE(9) MGB_CASE(10)
#undef MGB_CASE
        default: throw InvalidArgument("launch_dense_eval: nD out of range");
    }
    if (mode == MODE_F1) {
        // ret_a = sum_{k : state(k) = a} D_k' Y_k   (src/convex.jl:174-177)
        for (int a = 0; a < P.nu; ++a) {
            double* ra = P.out_ret + (int64_t)a * n;
            bool first = true;
            for (int k = 0; k < P.nD; ++k) {
                if (P.D_state[k] != a) continue;
                const double* Yk = P.dn_Y + (int64_t)k * n;
                const double* op = P.ops[P.D_op[k]];
                if (op == nullptr) {
                    if (first) MGB_HIP_CHECK(hipMemcpyAsync(ra, Yk, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
                    else launch_axpy(1.0, Yk, ra, n, st);
                } else {
                    launch_dense_gemv_t(n, n, op, n, Yk, ra, !first, st);
                }
                first = false;
            }
            if (first) MGB_HIP_CHECK(hipMemsetAsync(ra, 0, sizeof(double) * n, st));
        }
    }
}

void launch_dense_lu_solve(int m, const int32_t* Hptr, const int32_t* Hcol, const double* Hval, double* scratch_mm, const double* g,
                           double* x, int32_t* status, hipStream_t st) {
    hipLaunchKernelGGL(dense_lu_solve_kernel, dim3(1), dim3(LU_THREADS), 0, st, m, Hptr, Hcol, Hval, scratch_mm, g, x, status);
    MGB_HIP_CHECK(hipGetLastError());
}

}  // namespace mgbhip
