// dense.hpp -- dense-operator (spectral) twin of the element kernels.
//
// Spectral geometries carry ONE notional element whose operators are dense n x n matrices
// (reference: src/spectral1d.jl:100-108, src/spectral2d.jl); the reference evaluates the
// same Barrier closures on them with dense BLAS (Matrix{T} * Vector{T}, R' * (D' * diag * D) * R,
// src/convex.jl:155-202) and a dense Cholesky.  Here a problem with p > 64 nodes in its single
// element takes this path: GEMVs for D*z and D'*Y, a node-parallel cone kernel, and the Hessian
//     H = (D R)' * Ybar * (D R),   Ybar = the nD x nD block matrix of diagonal weights,
// as one symmetric fp64 MFMA GEMM  H = DR' * W  with  W = Ybar * DR  (v_mfma_f64_16x16x4_f64).
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace mgbhip {

// y = A*x          (A rows x cols, column-major, leading dimension lda)
void launch_dense_gemv_n(int rows, int cols, const double* A, int64_t lda, const double* x, double* y,
                         hipStream_t st);
// y (+)= A'*x
void launch_dense_gemv_t(int rows, int cols, const double* A, int64_t lda, const double* x, double* y, bool add,
                         hipStream_t st);
// C (M x N, ldc) (+)= A' * diag(v) * B    A: K x M (lda), B: K x N (ldb), v: K or nullptr.
// symmetric: M == N and the product is known to be symmetric; only tiles on or above the
// block diagonal are computed and each off-diagonal tile is mirrored.
void launch_dense_gemm_tn(int M, int N, int K, const double* A, int64_t lda, const double* v, const double* B,
                          int64_t ldb, double* C, int64_t ldc, bool accumulate, bool symmetric, hipStream_t st);
// At = A'   (n x n, column-major, contiguous)
void launch_dense_transpose(int n, const double* A, double* At, hipStream_t st);
// W[k*n + i, j] = sum_k' Y[i][k,k'] * DR[k'*n + i, j]   for klo <= k <= khi  (Y: n x NT upper-triangle rows)
void launch_dense_weight(int NY, int klo, int khi, int64_t n, int64_t m, int64_t ld, const double* DR,
                         const double* Yh, double* W, hipStream_t st);
// node-wise evaluation (same modes and outputs as launch_elem) for the dense path
void launch_dense_eval(const ElemParams& P, int mode, hipStream_t st);
int64_t dense_grid(int64_t n);
// Pivoted fallback of `solve(symmetric(H), g)` for small systems (src/utils.jl:142-145: Cholesky -> LDL' -> LU): dense LU with
// partial pivoting of the matrix whose upper triangle is the CSR H's; x = H^{-1} g; *status |= 1 when singular / non-finite.
// scratch_mm: m*m doubles.  x must not alias g.
constexpr int DENSE_LU_MAX_M = 2048;
void launch_dense_lu_solve(int m, const int32_t* Hptr, const int32_t* Hcol, const double* Hval, double* scratch_mm, const double* g,
                           double* x, int32_t* status, hipStream_t st);

}  // namespace mgbhip
