// mf_host_check.cpp -- TEST INFRASTRUCTURE (not shipped, not linked into libmgbhip.so).
// Runs the numeric phase of the multifrontal Cholesky on the host, following exactly the
// plan produced by csrc/mf_analysis.cpp (same scatter lists, relative indices, level
// order as the device kernels in csrc/mf_numeric.hip), so the symbolic analysis can be
// validated on a machine without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../multigridbarrier.jl_amd/csrc/mf_analysis.hpp"

using namespace mgbhip;

extern "C" int mf_host_solve(int64_t n, const int32_t* rowptr, const int32_t* colidx,
                             const double* values, const double* b, double* x, int32_t leaf_size,
                             double* stats /* 8 */) {
    MfPlan plan;
    MfOptions opt;
    if (leaf_size > 0) opt.leaf_size = leaf_size;
    try {
        mf_analyze(n, rowptr, colidx, opt, plan);
    } catch (const std::exception& e) {
        return -1;
    }
    std::vector<double> arena((size_t)plan.arena_doubles, 0.0);
    std::vector<double> uvec((size_t)plan.uvec_doubles, 0.0);
    int status = 0;
    const int32_t nf = (int32_t)plan.fronts.size();
    // factor: fronts are sorted by level, children strictly earlier
    for (int32_t i = 0; i < nf; ++i) {
        const Front& f = plan.fronts[i];
        double* F = arena.data() + f.F_off;
        const int32_t m = f.m, k = f.k;
        for (int32_t t = 0; t < f.a_cnt; ++t) F[plan.a_dst[f.a_off + t]] = values[plan.a_src[f.a_off + t]];
        for (int32_t c = 0; c < f.nchild; ++c) {
            const Front& ch = plan.fronts[plan.children[f.child_off + c]];
            const double* U = arena.data() + ch.F_off;
            const int32_t mc = ch.m, kc = ch.k;
            const int32_t* rel = plan.rel.data() + ch.rel_off;
            for (int32_t j = kc; j < mc; ++j)
                for (int32_t r = j; r < mc; ++r) F[rel[r - kc] + (int64_t)rel[j - kc] * m] += U[r + (int64_t)j * mc];
        }
        for (int32_t j = 0; j < k; ++j) {
            double d = F[j + (int64_t)j * m];
            if (!(d > 0)) status = 3;
            double l = std::sqrt(d);
            F[j + (int64_t)j * m] = l;
            for (int32_t r = j + 1; r < m; ++r) F[r + (int64_t)j * m] /= l;
            for (int32_t c2 = j + 1; c2 < m; ++c2) {
                double lc = F[c2 + (int64_t)j * m];
                for (int32_t r = c2; r < m; ++r) F[r + (int64_t)c2 * m] -= F[r + (int64_t)j * m] * lc;
            }
        }
    }
    // forward
    std::vector<double> y((size_t)n, 0.0), t;
    for (int32_t i = 0; i < nf; ++i) {
        const Front& f = plan.fronts[i];
        const double* F = arena.data() + f.F_off;
        const int32_t* idx = plan.front_idx.data() + f.idx_off;
        const int32_t m = f.m, k = f.k;
        t.assign(m, 0.0);
        for (int32_t j = 0; j < k; ++j) t[j] = b[idx[j]];
        for (int32_t c = 0; c < f.nchild; ++c) {
            const Front& ch = plan.fronts[plan.children[f.child_off + c]];
            const int32_t* rel = plan.rel.data() + ch.rel_off;
            for (int32_t j = 0; j < ch.m - ch.k; ++j) t[rel[j]] += uvec[ch.u_off + j];
        }
        for (int32_t j = 0; j < k; ++j) {
            t[j] /= F[j + (int64_t)j * m];
            for (int32_t r = j + 1; r < m; ++r) t[r] -= F[r + (int64_t)j * m] * t[j];
        }
        for (int32_t j = 0; j < k; ++j) y[idx[j]] = t[j];
        for (int32_t j = k; j < m; ++j) uvec[f.u_off + j - k] = t[j];
    }
    // backward (roots first)
    for (int32_t i = nf - 1; i >= 0; --i) {
        const Front& f = plan.fronts[i];
        const double* F = arena.data() + f.F_off;
        const int32_t* idx = plan.front_idx.data() + f.idx_off;
        const int32_t m = f.m, k = f.k;
        t.assign(m, 0.0);
        for (int32_t j = 0; j < k; ++j) t[j] = y[idx[j]];
        for (int32_t j = k; j < m; ++j) t[j] = x[idx[j]];
        for (int32_t j = k - 1; j >= 0; --j) {
            double s = t[j];
            for (int32_t r = j + 1; r < m; ++r) s -= F[r + (int64_t)j * m] * t[r];
            t[j] = s / F[j + (int64_t)j * m];
        }
        for (int32_t j = 0; j < k; ++j) x[idx[j]] = t[j];
    }
    if (stats) {
        stats[0] = (double)nf;
        stats[1] = (double)plan.max_m;
        stats[2] = (double)plan.arena_doubles;
        stats[3] = (double)plan.factor_flops;
        stats[4] = (double)plan.peeled;
        stats[5] = (double)plan.peel_rounds;
        stats[6] = (double)(plan.level_ptr.size() - 1);
        stats[7] = (double)plan.uvec_doubles;
    }
    return status;
}
