// mf_numeric.hip -- numeric phase of the multifrontal Cholesky on gfx950.
//
// One workgroup per frontal matrix, all fronts of one elimination-tree level and size
// class in one launch (leaves first).  Small fronts (m <= lds_cap) are assembled,
// factored and written back out of LDS; larger fronts work in their HBM arena slot.
// Every extend-add runs child by child in a fixed order inside the parent's workgroup:
// no atomics, bitwise reproducible factors.
//
// Frontal layout: column-major m x m, ld = m; columns [0,k) are the L panel after the
// factorization, the trailing (m-k)^2 lower triangle is the update matrix the parent reads.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/mgbhip.h"
#include "mf_solver.hpp"

namespace mgbhip {

namespace {

constexpr int TX = 16;   // row lanes
constexpr int TY = 16;   // column lanes (TX*TY = 256 threads)

template <bool USE_LDS>
__global__ __launch_bounds__(256) void mf_factor_kernel(const FrontDev* __restrict__ fr, int32_t first,
                                                        const int32_t* __restrict__ children,
                                                        const int32_t* __restrict__ rel,
                                                        const int32_t* __restrict__ a_src,
                                                        const int32_t* __restrict__ a_dst,
                                                        const double* __restrict__ Hval,
                                                        double* __restrict__ arena,
                                                        int32_t* __restrict__ status) {
    extern __shared__ double sh[];
    const FrontDev F = fr[first + blockIdx.x];
    double* Fg = arena + F.F_off;
    double* W = USE_LDS ? sh : Fg;
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x;
    const int tx = tid % TX, ty = tid / TX;
    const int64_t mm = (int64_t)m * m;

    for (int64_t i = tid; i < mm; i += 256) W[i] = 0.0;
    __syncthreads();
    for (int t = tid; t < F.a_cnt; t += 256) W[a_dst[F.a_off + t]] = Hval[a_src[F.a_off + t]];
    __syncthreads();
    for (int c = 0; c < F.nchild; ++c) {
        const FrontDev C = fr[children[F.child_off + c]];
        const double* U = arena + C.F_off;
        const int mc = C.m, kc = C.k, b = mc - kc;
        const int32_t* rl = rel + C.rel_off;
        for (int j = ty; j < b; j += TY) {
            const int64_t dcol = (int64_t)rl[j] * m;
            const double* Uc = U + (int64_t)(kc + j) * mc + kc;
            for (int r = j + tx; r < b; r += TX) W[rl[r] + dcol] += Uc[r];
        }
        __syncthreads();
    }
    bool bad = false;
    for (int j = 0; j < k; ++j) {
        const double d = W[j + (int64_t)j * m];
        if (!(d > 0.0)) bad = true;
        const double l = sqrt(d);
        const double inv = 1.0 / l;
        __syncthreads();
        for (int r = j + tid; r < m; r += 256) {
            const int64_t a = r + (int64_t)j * m;
            W[a] = (r == j) ? l : W[a] * inv;
        }
        __syncthreads();
        const double* Lj = W + (int64_t)j * m;
        for (int c2 = j + 1 + ty; c2 < m; c2 += TY) {
            const double lc = Lj[c2];
            double* Wc = W + (int64_t)c2 * m;
            for (int r = c2 + tx; r < m; r += TX) Wc[r] -= Lj[r] * lc;
        }
        __syncthreads();
    }
    if (bad && tid == 0) atomicOr(status, 1);
    if (USE_LDS) {
        for (int64_t i = tid; i < mm; i += 256) Fg[i] = W[i];
    }
}

// forward substitution: t = L^{-1} (b + children's updates); y[piv] = t[0:k]; u = t[k:m]
__global__ __launch_bounds__(256) void mf_forward_kernel(const FrontDev* __restrict__ fr, int32_t first,
                                                         const int32_t* __restrict__ front_idx,
                                                         const int32_t* __restrict__ children,
                                                         const int32_t* __restrict__ rel,
                                                         const double* __restrict__ arena,
                                                         const double* __restrict__ b,
                                                         double* __restrict__ y, double* __restrict__ uvec,
                                                         double* __restrict__ tglobal) {
    extern __shared__ double sh[];
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x;
    double* t = tglobal ? tglobal + F.idx_off : sh;   // big fronts: scratch indexed like front_idx
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
    for (int j = 0; j < k; ++j) {
        const double tj = t[j] / Fm[j + (int64_t)j * m];
        __syncthreads();
        if (tid == 0) t[j] = tj;
        const double* Lj = Fm + (int64_t)j * m;
        for (int r = j + 1 + tid; r < m; r += 256) t[r] -= Lj[r] * tj;
        __syncthreads();
    }
    for (int j = tid; j < m; j += 256) {
        if (j < k) y[idx[j]] = t[j];
        else uvec[F.u_off + j - k] = t[j];
    }
}

// backward substitution: x[piv] = L11^{-T} (y[piv] - L21^T x[bnd])
__global__ __launch_bounds__(256) void mf_backward_kernel(const FrontDev* __restrict__ fr, int32_t first,
                                                          const int32_t* __restrict__ front_idx,
                                                          const double* __restrict__ arena,
                                                          const double* __restrict__ y,
                                                          double* __restrict__ x,
                                                          double* __restrict__ tglobal) {
    extern __shared__ double sh[];
    __shared__ double red[256];
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x;
    double* t = tglobal ? tglobal + F.idx_off : sh;
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    for (int j = tid; j < m; j += 256) t[j] = (j < k) ? y[idx[j]] : x[idx[j]];
    __syncthreads();
    for (int j = k - 1; j >= 0; --j) {
        const double* Lj = Fm + (int64_t)j * m;
        double s = 0.0;
        for (int r = j + 1 + tid; r < m; r += 256) s += Lj[r] * t[r];
        red[tid] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) red[tid] += red[tid + off];
            __syncthreads();
        }
        if (tid == 0) t[j] = (t[j] - red[0]) / Lj[j];
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
    d_status.alloc(1);
    d_status.zero(st);

    // dynamic LDS above 64 KB needs an explicit opt-in; fall back to the 64 KB classes if refused
    lds_cap = 88;
    if (hipFuncSetAttribute((const void*)mf_factor_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            128 * 128 * 8) == hipSuccess)
        lds_cap = 128;
    else
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
            if (cls) {
                while (j < end && plan.fronts[j].m <= cls) ++j;
            } else {
                j = end;   // sorted by m: everything left in the level is large
            }
            level_launches[l].push_back(MfLaunch{i, j - i, cls});
            i = j;
        }
    }
    analyzed = true;
    MGB_HIP_CHECK(hipStreamSynchronize(st));   // host staging vectors go out of scope
}

void MfSolver::factor(const double* d_values, hipStream_t st, StageTimers* timers) {
    MGB_REQUIRE(analyzed, "MfSolver::factor before analyze");
    if (timers) timers->begin("factor");
    d_status.zero(st);
    for (auto& lev : level_launches)
        for (auto& L : lev) {
            if (L.count == 0) continue;
            if (L.cls) {
                size_t lds = (size_t)L.cls * L.cls * sizeof(double);
                hipLaunchKernelGGL(mf_factor_kernel<true>, dim3(L.count), dim3(256), lds, st, d_fronts.p, L.first,
                                   d_children.p, d_rel.p, d_a_src.p, d_a_dst.p, d_values, d_arena.p, d_status.p);
            } else {
                hipLaunchKernelGGL(mf_factor_kernel<false>, dim3(L.count), dim3(256), 0, st, d_fronts.p, L.first,
                                   d_children.p, d_rel.p, d_a_src.p, d_a_dst.p, d_values, d_arena.p, d_status.p);
            }
        }
    MGB_HIP_CHECK(hipGetLastError());
    if (timers) timers->end();
}

void MfSolver::solve(const double* d_b, double* d_x, hipStream_t st, StageTimers* timers) {
    MGB_REQUIRE(analyzed, "MfSolver::solve before analyze");
    if (timers) timers->begin("trisolve");
    // work vectors of fronts too large for LDS live in a scratch array indexed like front_idx
    if (!d_tbig_ready) {
        d_tbig.alloc(plan.front_idx.size() ? plan.front_idx.size() : 1);
        d_tbig_ready = true;
    }
    const size_t LDS_T_CAP = 4096;   // doubles
    for (auto& lev : level_launches)
        for (auto& L : lev) {
            if (L.count == 0) continue;
            int32_t mmax = plan.fronts[L.first + L.count - 1].m;
            bool big = (size_t)mmax > LDS_T_CAP;
            size_t lds = big ? 0 : (size_t)mmax * sizeof(double);
            hipLaunchKernelGGL(mf_forward_kernel, dim3(L.count), dim3(256), lds, st, d_fronts.p, L.first,
                               d_front_idx.p, d_children.p, d_rel.p, d_arena.p, d_b, d_y.p, d_uvec.p,
                               big ? d_tbig.p : (double*)nullptr);
        }
    for (int32_t l = (int32_t)level_launches.size() - 1; l >= 0; --l)
        for (auto it = level_launches[l].rbegin(); it != level_launches[l].rend(); ++it) {
            const MfLaunch& L = *it;
            if (L.count == 0) continue;
            int32_t mmax = plan.fronts[L.first + L.count - 1].m;
            bool big = (size_t)mmax > LDS_T_CAP;
            size_t lds = big ? 0 : (size_t)mmax * sizeof(double);
            hipLaunchKernelGGL(mf_backward_kernel, dim3(L.count), dim3(256), lds, st, d_fronts.p, L.first,
                               d_front_idx.p, d_arena.p, d_y.p, d_x, big ? d_tbig.p : (double*)nullptr);
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
