// mf_numeric.hip -- numeric phase of the multifrontal LDL' factorization on gfx950.
//
// `solve(symmetric(H), g)` in the reference is CHOLMOD's Cholesky with an LDL' fallback when
// H is not numerically positive definite (Julia's `\` for Symmetric sparse matrices,
// src/utils.jl:142-145).  One square-root-free LDL' (fixed ordering, no pivoting -- like
// CHOLMOD's) covers both: for SPD input it is the Cholesky factor column-scaled, for a
// numerically indefinite H it still returns the direction the reference would get, and the
// Newton loop's lambda^2 <= 0 test (src/newton.jl:257-271) decides.  Only an exactly zero or
// non-finite pivot is an error.
//
// Frontal layout: column-major m x m, ld = m.  After factorization columns [0,k) hold the
// strictly lower part of the unit-lower L panel with D on the diagonal; the trailing (m-k)^2
// lower triangle is the update matrix the parent reads.
//
// Small fronts (m <= lds_cap) are assembled, factored and written back out of LDS by one
// workgroup each, all fronts of one tree level and size class in one launch.  Large fronts
// are processed by a batch of multi-workgroup kernels per level: column-tiled assembly, then
// per 32-column panel a (redundant diagonal LDL' + row-tile triangular solve) kernel and a
// 64x64-tiled symmetric rank-32 update kernel.  Every extend-add runs child by child in a
// fixed order on disjoint destination columns: no atomics, bitwise reproducible factors.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/mgbhip.h"
#include "mf_solver.hpp"

namespace mgbhip {

namespace {

constexpr int TX = 16;    // row lanes of the 2-D thread maps
constexpr int NB = 32;    // panel width of the large-front path
constexpr int CT = 32;    // destination columns per workgroup in the large-front assembly
constexpr int TR = 256;   // rows per workgroup in the panel solve
constexpr int ST = 64;    // tile edge of the symmetric update

// ------------------------------------------------------------------------------------------------
// small fronts
// ------------------------------------------------------------------------------------------------

__global__ void mf_factor_small(const FrontDev* __restrict__ fr, int32_t first,
                                const int32_t* __restrict__ children, const int32_t* __restrict__ rel,
                                const int32_t* __restrict__ a_src, const int32_t* __restrict__ a_dst,
                                const double* __restrict__ Hval, double* __restrict__ arena,
                                int32_t* __restrict__ status) {
    extern __shared__ double W[];
    const FrontDev F = fr[first + blockIdx.x];
    double* Fg = arena + F.F_off;
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int tx = tid % TX, ty = tid / TX, TYn = nt / TX;
    const int mm = m * m;

    for (int i = tid; i < mm; i += nt) W[i] = 0.0;
    __syncthreads();
    for (int t = tid; t < F.a_cnt; t += nt) W[a_dst[F.a_off + t]] = Hval[a_src[F.a_off + t]];
    __syncthreads();
    for (int c = 0; c < F.nchild; ++c) {
        const FrontDev C = fr[children[F.child_off + c]];
        const double* U = arena + C.F_off;
        const int mc = C.m, kc = C.k, b = mc - kc;
        const int32_t* rl = rel + C.rel_off;
        for (int j = ty; j < b; j += TYn) {
            const int dcol = rl[j] * m;
            const double* Uc = U + (int64_t)(kc + j) * mc + kc;
            for (int r = j + tx; r < b; r += TX) W[rl[r] + dcol] += Uc[r];
        }
        __syncthreads();
    }
    bool bad = false;
    for (int j = 0; j < k; ++j) {
        const double d = W[j + j * m];
        if (d == 0.0 || !isfinite(d)) bad = true;
        const double inv = 1.0 / d;
        // trailing update with the unscaled column, then scale the column: L[r,j] = W[r,j] / d
        double* Lj = W + j * m;
        for (int c2 = j + 1 + ty; c2 < m; c2 += TYn) {
            const double lc = Lj[c2] * inv;
            double* Wc = W + c2 * m;
            for (int r = c2 + tx; r < m; r += TX) Wc[r] -= Lj[r] * lc;
        }
        __syncthreads();
        for (int r = j + 1 + tid; r < m; r += nt) Lj[r] *= inv;
        __syncthreads();
    }
    if (bad && tid == 0) atomicOr(status, 1);
    for (int i = tid; i < mm; i += nt) Fg[i] = W[i];
}

// forward: t = L^{-1}(b + children's updates); y[piv] = D^{-1} t[0:k]; u = t[k:m]
__global__ void mf_forward_small(const FrontDev* __restrict__ fr, int32_t first,
                                 const int32_t* __restrict__ front_idx, const int32_t* __restrict__ children,
                                 const int32_t* __restrict__ rel, const double* __restrict__ arena,
                                 const double* __restrict__ b, double* __restrict__ y, double* __restrict__ uvec) {
    extern __shared__ double t[];
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    for (int j = tid; j < m; j += nt) t[j] = (j < k) ? b[idx[j]] : 0.0;
    __syncthreads();
    for (int c = 0; c < F.nchild; ++c) {
        const FrontDev C = fr[children[F.child_off + c]];
        const int32_t* rl = rel + C.rel_off;
        const double* uc = uvec + C.u_off;
        for (int j = tid; j < C.m - C.k; j += nt) t[rl[j]] += uc[j];
        __syncthreads();
    }
    for (int j = 0; j < k; ++j) {
        const double tj = t[j];
        const double* Lj = Fm + (int64_t)j * m;
        for (int r = j + 1 + tid; r < m; r += nt) t[r] -= Lj[r] * tj;
        __syncthreads();
    }
    for (int j = tid; j < m; j += nt) {
        if (j < k) y[idx[j]] = t[j] / Fm[j + (int64_t)j * m];
        else uvec[F.u_off + j - k] = t[j];
    }
}

// backward: x[piv] = L11^{-T} (y[piv] - L21^T x[bnd])
__global__ void mf_backward_small(const FrontDev* __restrict__ fr, int32_t first,
                                  const int32_t* __restrict__ front_idx, const double* __restrict__ arena,
                                  const double* __restrict__ y, double* __restrict__ x) {
    extern __shared__ double t[];
    __shared__ double red[256];
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    for (int j = tid; j < m; j += nt) t[j] = (j < k) ? y[idx[j]] : x[idx[j]];
    __syncthreads();
    for (int j = k - 1; j >= 0; --j) {
        const double* Lj = Fm + (int64_t)j * m;
        double s = 0.0;
        for (int r = j + 1 + tid; r < m; r += nt) s += Lj[r] * t[r];
        red[tid] = s;
        __syncthreads();
        for (int off = nt >> 1; off > 0; off >>= 1) {
            if (tid < off) red[tid] += red[tid + off];
            __syncthreads();
        }
        if (tid == 0) t[j] -= red[0];
        __syncthreads();
    }
    for (int j = tid; j < k; j += nt) x[idx[j]] = t[j];
}

// ------------------------------------------------------------------------------------------------
// large fronts: batched multi-workgroup kernels (grid.y = front within the batch)
// ------------------------------------------------------------------------------------------------

// Assembly of destination columns [c0, c0 + CT): zero, scatter A, extend-add the children.
__global__ __launch_bounds__(256) void mf_big_assemble(const FrontDev* __restrict__ fr, int32_t first,
                                                       const int32_t* __restrict__ children,
                                                       const int32_t* __restrict__ rel,
                                                       const int32_t* __restrict__ a_src,
                                                       const int32_t* __restrict__ a_dst,
                                                       const double* __restrict__ Hval, double* __restrict__ arena) {
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m;
    const int c0 = blockIdx.x * CT;
    if (c0 >= m) return;
    const int c1 = min(c0 + CT, m);
    double* W = arena + F.F_off;
    const int tid = threadIdx.x;
    const int tx = tid % TX, ty = tid / TX;     // 16 x 16
    for (int c = c0 + ty; c < c1; c += 16) {
        double* Wc = W + (int64_t)c * m;
        for (int r = c + tx; r < m; r += TX) Wc[r] = 0.0;
    }
    __syncthreads();
    {   // A entries are ordered by destination column: binary search the range of [c0, c1)
        const int32_t* ad = a_dst + F.a_off;
        int lo = 0, hi = F.a_cnt;
        const int64_t key0 = (int64_t)c0 * m, key1 = (int64_t)c1 * m;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (ad[mid] < key0) lo = mid + 1; else hi = mid; }
        int beg = lo;
        hi = F.a_cnt;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (ad[mid] < key1) lo = mid + 1; else hi = mid; }
        for (int t = beg + tid; t < lo; t += 256) W[ad[t]] = Hval[a_src[F.a_off + t]];
    }
    __syncthreads();
    for (int c = 0; c < F.nchild; ++c) {
        const FrontDev C = fr[children[F.child_off + c]];
        const double* U = arena + C.F_off;
        const int mc = C.m, kc = C.k, b = mc - kc;
        const int32_t* rl = rel + C.rel_off;
        // child columns whose destination lies in [c0, c1): rel is increasing
        int lo = 0, hi = b;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (rl[mid] < c0) lo = mid + 1; else hi = mid; }
        const int jb = lo;
        hi = b;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (rl[mid] < c1) lo = mid + 1; else hi = mid; }
        const int je = lo;
        for (int j = jb + ty; j < je; j += 16) {
            double* Wc = W + (int64_t)rl[j] * m;
            const double* Uc = U + (int64_t)(kc + j) * mc + kc;
            for (int r = j + tx; r < b; r += TX) Wc[rl[r]] += Uc[r];
        }
        __syncthreads();
    }
}

// Diagonal step: one workgroup per front factors the nb x nb diagonal block (LDL') in LDS.
__global__ __launch_bounds__(256) void mf_big_diag(const FrontDev* __restrict__ fr, int32_t first, int j0,
                                                   double* __restrict__ arena, int32_t* __restrict__ status) {
    __shared__ double Dk[NB][NB + 1];
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    if (j0 >= k) return;
    const int nb = min(NB, k - j0);
    double* W = arena + F.F_off;
    const int tid = threadIdx.x;
    for (int i = tid; i < nb * nb; i += 256) {
        const int r = i % nb, c = i / nb;
        Dk[r][c] = (r >= c) ? W[(j0 + r) + (int64_t)(j0 + c) * m] : 0.0;
    }
    __syncthreads();
    bool bad = false;
    for (int j = 0; j < nb; ++j) {
        const double d = Dk[j][j];
        if (d == 0.0 || !isfinite(d)) bad = true;
        const double inv = 1.0 / d;
        for (int i = tid; i < nb * nb; i += 256) {       // trailing update with the unscaled column
            const int r = i % nb, c = i / nb;
            if (c > j && r >= c) Dk[r][c] -= Dk[r][j] * (Dk[c][j] * inv);
        }
        __syncthreads();
        if (tid > j && tid < nb) Dk[tid][j] *= inv;
        __syncthreads();
    }
    if (bad && tid == 0) atomicOr(status, 1);
    for (int i = tid; i < nb * nb; i += 256) {
        const int r = i % nb, c = i / nb;
        if (r >= c) W[(j0 + r) + (int64_t)(j0 + c) * m] = Dk[r][c];
    }
}

// Panel step: each workgroup loads the factored diagonal block and solves its TR rows of the
// panel: L21 = A21 L11^{-T} D^{-1}.
__global__ __launch_bounds__(256) void mf_big_panel(const FrontDev* __restrict__ fr, int32_t first, int j0,
                                                    double* __restrict__ arena) {
    __shared__ double Dk[NB][NB + 1];
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m, k = F.k;
    if (j0 >= k) return;
    const int nb = min(NB, k - j0);
    const int r0 = j0 + nb + blockIdx.x * TR;
    if (r0 >= m) return;
    double* W = arena + F.F_off;
    const int tid = threadIdx.x;
    for (int i = tid; i < nb * nb; i += 256) {
        const int r = i % nb, c = i / nb;
        Dk[r][c] = (r >= c) ? W[(j0 + r) + (int64_t)(j0 + c) * m] : 0.0;
    }
    __syncthreads();
    const int r = r0 + tid;
    if (r < m) {
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = (c < nb) ? W[r + (int64_t)(j0 + c) * m] : 0.0;
        // y L11' = a  (unit lower L11), then l = y / d
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            if (c < nb) {
                double v = a[c];
#pragma unroll
                for (int q = 0; q < NB; ++q)
                    if (q < c) v -= a[q] * Dk[c][q];
                a[c] = v;
            }
        }
#pragma unroll
        for (int c = 0; c < NB; ++c)
            if (c < nb) W[r + (int64_t)(j0 + c) * m] = a[c] / Dk[c][c];
    }
}

// Symmetric update of the trailing block with the finished panel:
// C[r, c] -= sum_q L[r, q] d_q L[c, q], 64 x 64 tiles of the lower triangle, 4 x 4 per thread.
__global__ __launch_bounds__(256) void mf_big_update(const FrontDev* __restrict__ fr, int32_t first, int j0,
                                                     double* __restrict__ arena) {
    __shared__ double Pi[NB][ST + 1];
    __shared__ double Qj[NB][ST + 1];
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m, k = F.k;
    if (j0 >= k) return;
    const int nb = min(NB, k - j0);
    const int j1 = j0 + nb;
    const int T = (m - j1 + ST - 1) / ST;
    // decode the tile pair (ti >= tj) from the linear index
    const int lin = blockIdx.x;
    int ti = (int)((sqrt(8.0 * lin + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= lin) ++ti;
    while (ti * (ti + 1) / 2 > lin) --ti;
    const int tj = lin - ti * (ti + 1) / 2;
    if (ti >= T) return;
    double* W = arena + F.F_off;
    const int tid = threadIdx.x;
    const int rbase = j1 + ti * ST, cbase = j1 + tj * ST;
    for (int i = tid; i < nb * ST; i += 256) {
        const int rr = i % ST, q = i / ST;
        const int r = rbase + rr, c = cbase + rr;
        Pi[q][rr] = (r < m) ? W[r + (int64_t)(j0 + q) * m] : 0.0;
        Qj[q][rr] = (c < m) ? W[c + (int64_t)(j0 + q) * m] * W[(j0 + q) + (int64_t)(j0 + q) * m] : 0.0;
    }
    __syncthreads();
    const int tx = tid % 16, ty = tid / 16;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int q = 0; q < nb; ++q) {
        double pr[4], qc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) pr[a] = Pi[q][tx + 16 * a];
#pragma unroll
        for (int b = 0; b < 4; ++b) qc[b] = Qj[q][ty + 16 * b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] += pr[a] * qc[b];
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int c = cbase + ty + 16 * b;
        if (c >= m) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = rbase + tx + 16 * a;
            if (r < m && r >= c) W[r + (int64_t)c * m] -= acc[a][b];
        }
    }
}

// Blocked forward substitution for a large front (one workgroup per front).
__global__ __launch_bounds__(256) void mf_forward_big(const FrontDev* __restrict__ fr, int32_t first,
                                                      const int32_t* __restrict__ front_idx,
                                                      const int32_t* __restrict__ children,
                                                      const int32_t* __restrict__ rel,
                                                      const double* __restrict__ arena,
                                                      const double* __restrict__ b, double* __restrict__ y,
                                                      double* __restrict__ uvec, double* __restrict__ tglobal) {
    extern __shared__ double sh[];
    __shared__ double Dk[NB][NB + 1];
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x;
    double* t = tglobal ? tglobal + F.idx_off : sh;
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    for (int j = tid; j < m; j += 256) t[j] = (j < k) ? b[idx[j]] : 0.0;
    __syncthreads();
    for (int c = 0; c < F.nchild; ++c) {
        const FrontDev C = fr[children[F.child_off + c]];
        const int32_t* rl = rel + C.rel_off;
        const double* uc = uvec + C.u_off;
        for (int j = tid; j < C.m - C.k; j += 256) t[rl[j]] += uc[j];
        __syncthreads();
    }
    for (int j0 = 0; j0 < k; j0 += NB) {
        const int nb = min(NB, k - j0);
        for (int i = tid; i < nb * nb; i += 256) {
            const int r = i % nb, c = i / nb;
            Dk[r][c] = (r > c) ? Fm[(j0 + r) + (int64_t)(j0 + c) * m] : 0.0;
        }
        __syncthreads();
        if (tid < 64) {                       // wave 0: unit-lower solve of the diagonal block in registers
            double v = (tid < nb) ? t[j0 + tid] : 0.0;
            for (int c = 0; c < nb; ++c) {
                const double tc = __shfl(v, c, 64);
                if (tid > c && tid < nb) v -= Dk[tid][c] * tc;
            }
            if (tid < nb) t[j0 + tid] = v;
        }
        __syncthreads();
        for (int r = j0 + nb + tid; r < m; r += 256) {
            double v = t[r];
            for (int c = 0; c < nb; ++c) v -= Fm[r + (int64_t)(j0 + c) * m] * t[j0 + c];
            t[r] = v;
        }
        __syncthreads();
    }
    for (int j = tid; j < m; j += 256) {
        if (j < k) y[idx[j]] = t[j] / Fm[j + (int64_t)j * m];
        else uvec[F.u_off + j - k] = t[j];
    }
}

// Blocked backward substitution for a large front.
__global__ __launch_bounds__(256) void mf_backward_big(const FrontDev* __restrict__ fr, int32_t first,
                                                       const int32_t* __restrict__ front_idx,
                                                       const double* __restrict__ arena,
                                                       const double* __restrict__ y, double* __restrict__ x,
                                                       double* __restrict__ tglobal) {
    extern __shared__ double sh[];
    __shared__ double Dk[NB][NB + 1];
    __shared__ double part[4][NB];
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    double* t = tglobal ? tglobal + F.idx_off : sh;
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    for (int j = tid; j < m; j += 256) t[j] = (j < k) ? y[idx[j]] : x[idx[j]];
    __syncthreads();
    const int nblk = (k + NB - 1) / NB;
    for (int bi = nblk - 1; bi >= 0; --bi) {
        const int j0 = bi * NB;
        const int nb = min(NB, k - j0);
        // acc[c] = sum_{r >= j0+nb} L[r, j0+c] t[r]
        double acc[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) acc[c] = 0.0;
        for (int r = j0 + nb + tid; r < m; r += 256) {
            const double tr = t[r];
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if (c < nb) acc[c] += Fm[r + (int64_t)(j0 + c) * m] * tr;
        }
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            double v = acc[c];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) part[wave][c] = v;
        }
        for (int i = tid; i < nb * nb; i += 256) {
            const int r = i % nb, c = i / nb;
            Dk[r][c] = (r > c) ? Fm[(j0 + r) + (int64_t)(j0 + c) * m] : 0.0;
        }
        __syncthreads();
        if (tid < 64) {                       // wave 0: unit-upper (L11') solve in registers
            double v = 0.0;
            if (tid < nb) v = t[j0 + tid] - (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]);
            for (int c = nb - 1; c >= 0; --c) {
                const double xc = __shfl(v, c, 64);
                if (tid < c) v -= Dk[c][tid] * xc;
            }
            if (tid < nb) t[j0 + tid] = v;
        }
        __syncthreads();
    }
    for (int j = tid; j < k; j += 256) x[idx[j]] = t[j];
}

}  // namespace

void MfSolver::analyze(int64_t n, const int32_t* rowptr, const int32_t* colidx, hipStream_t st) {
    MfOptions opt;
    mf_analyze(n, rowptr, colidx, opt, plan);
    const int32_t nf = (int32_t)plan.fronts.size();
    std::vector<FrontDev> fd(nf);
    for (int32_t i = 0; i < nf; ++i) {
        const Front& f = plan.fronts[i];
        fd[i] = FrontDev{f.k, f.m, f.nchild, f.a_cnt, f.F_off, f.idx_off, f.u_off, f.child_off, f.rel_off, f.a_off};
    }
    d_fronts.upload(fd, st);
    d_front_idx.upload(plan.front_idx, st);
    d_children.upload(plan.children, st);
    d_rel.upload(plan.rel, st);
    d_a_src.upload(plan.a_src, st);
    d_a_dst.upload(plan.a_dst, st);
    d_arena.alloc((size_t)std::max<int64_t>(plan.arena_doubles, 1));
    d_uvec.alloc((size_t)std::max<int64_t>(plan.uvec_doubles, 1));
    d_y.alloc((size_t)std::max<int64_t>(plan.n, 1));
    d_tbig.alloc(plan.front_idx.size() ? plan.front_idx.size() : 1);
    d_status.alloc(1);
    d_status.zero(st);

    // dynamic LDS above 64 KB needs an explicit opt-in; fall back to the 64 KB classes if refused
    lds_cap = 88;
    if (hipFuncSetAttribute((const void*)mf_factor_small, hipFuncAttributeMaxDynamicSharedMemorySize,
                            128 * 128 * 8) == hipSuccess)
        lds_cap = 128;
    else
        (void)hipGetLastError();
    (void)hipFuncSetAttribute((const void*)mf_forward_big, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)mf_backward_big, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipGetLastError();

    static const int32_t classes[] = {16, 32, 48, 64, 88, 128};
    level_launches.clear();
    const int32_t nlev = (int32_t)plan.level_ptr.size() - 1;
    level_launches.resize(nlev);
    for (int32_t l = 0; l < nlev; ++l) {
        int32_t i = plan.level_ptr[l];
        const int32_t end = plan.level_ptr[l + 1];
        while (i < end) {
            int32_t m = plan.fronts[i].m;
            int32_t cls = 0;
            for (int32_t c : classes)
                if (m <= c && c <= lds_cap) { cls = c; break; }
            int32_t j = i;
            MfLaunch L{};
            if (cls) {
                while (j < end && plan.fronts[j].m <= cls) ++j;
            } else {
                j = end;   // sorted by m: everything left in the level is large
            }
            L.first = i;
            L.count = j - i;
            L.cls = cls;
            L.max_m = plan.fronts[j - 1].m;
            L.max_k = 0;
            for (int32_t q = i; q < j; ++q) L.max_k = std::max(L.max_k, plan.fronts[q].k);
            level_launches[l].push_back(L);
            i = j;
        }
    }
    analyzed = true;
    MGB_HIP_CHECK(hipStreamSynchronize(st));   // host staging vectors go out of scope
}

static inline int small_threads(int cls) { return cls <= 16 ? 64 : (cls <= 32 ? 128 : 256); }

void MfSolver::factor(const double* d_values, hipStream_t st, StageTimers* timers) {
    MGB_REQUIRE(analyzed, "MfSolver::factor before analyze");
    if (timers) timers->begin("factor");
    d_status.zero(st);
    for (auto& lev : level_launches)
        for (auto& L : lev) {
            if (L.count == 0) continue;
            if (L.cls) {
                size_t lds = (size_t)L.cls * L.cls * sizeof(double);
                hipLaunchKernelGGL(mf_factor_small, dim3(L.count), dim3(small_threads(L.cls)), lds, st, d_fronts.p,
                                   L.first, d_children.p, d_rel.p, d_a_src.p, d_a_dst.p, d_values, d_arena.p,
                                   d_status.p);
            } else {
                const dim3 ga((L.max_m + CT - 1) / CT, L.count);
                hipLaunchKernelGGL(mf_big_assemble, ga, dim3(256), 0, st, d_fronts.p, L.first, d_children.p, d_rel.p,
                                   d_a_src.p, d_a_dst.p, d_values, d_arena.p);
                for (int j0 = 0; j0 < L.max_k; j0 += NB) {
                    const int rem = L.max_m - j0;                // rows below the panel start, at most
                    const dim3 gp(std::max(1, (rem + TR - 1) / TR), L.count);
                    hipLaunchKernelGGL(mf_big_diag, dim3(L.count), dim3(256), 0, st, d_fronts.p, L.first, j0, d_arena.p,
                                       d_status.p);
                    hipLaunchKernelGGL(mf_big_panel, gp, dim3(256), 0, st, d_fronts.p, L.first, j0, d_arena.p);
                    const int T = (rem - 1 + ST - 1) / ST;       // trailing tiles (upper bound)
                    if (T > 0) {
                        const dim3 gu(T * (T + 1) / 2, L.count);
                        hipLaunchKernelGGL(mf_big_update, gu, dim3(256), 0, st, d_fronts.p, L.first, j0, d_arena.p);
                    }
                }
            }
        }
    MGB_HIP_CHECK(hipGetLastError());
    if (timers) timers->end();
}

void MfSolver::solve(const double* d_b, double* d_x, hipStream_t st, StageTimers* timers) {
    MGB_REQUIRE(analyzed, "MfSolver::solve before analyze");
    if (timers) timers->begin("trisolve");
    const size_t LDS_T_CAP = 15000;   // doubles of work vector kept in LDS by the large-front solves
    for (auto& lev : level_launches)
        for (auto& L : lev) {
            if (L.count == 0) continue;
            if (L.cls) {
                hipLaunchKernelGGL(mf_forward_small, dim3(L.count), dim3(small_threads(L.cls)),
                                   (size_t)L.max_m * sizeof(double), st, d_fronts.p, L.first, d_front_idx.p,
                                   d_children.p, d_rel.p, d_arena.p, d_b, d_y.p, d_uvec.p);
            } else {
                const bool glob = (size_t)L.max_m > LDS_T_CAP;
                hipLaunchKernelGGL(mf_forward_big, dim3(L.count), dim3(256), glob ? 0 : (size_t)L.max_m * sizeof(double),
                                   st, d_fronts.p, L.first, d_front_idx.p, d_children.p, d_rel.p, d_arena.p, d_b,
                                   d_y.p, d_uvec.p, glob ? d_tbig.p : (double*)nullptr);
            }
        }
    for (int32_t l = (int32_t)level_launches.size() - 1; l >= 0; --l)
        for (auto it = level_launches[l].rbegin(); it != level_launches[l].rend(); ++it) {
            const MfLaunch& L = *it;
            if (L.count == 0) continue;
            if (L.cls) {
                hipLaunchKernelGGL(mf_backward_small, dim3(L.count), dim3(small_threads(L.cls)),
                                   (size_t)L.max_m * sizeof(double), st, d_fronts.p, L.first, d_front_idx.p,
                                   d_arena.p, d_y.p, d_x);
            } else {
                const bool glob = (size_t)L.max_m > LDS_T_CAP;
                hipLaunchKernelGGL(mf_backward_big, dim3(L.count), dim3(256),
                                   glob ? 0 : (size_t)L.max_m * sizeof(double), st, d_fronts.p, L.first,
                                   d_front_idx.p, d_arena.p, d_y.p, d_x, glob ? d_tbig.p : (double*)nullptr);
            }
        }
    MGB_HIP_CHECK(hipGetLastError());
    if (timers) timers->end();
}

int MfSolver::status(hipStream_t st) {
    int32_t h = 0;
    d_status.download(&h, 1, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    return h ? MGBHIP_ERR_NOT_SPD : MGBHIP_OK;
}

}  // namespace mgbhip
