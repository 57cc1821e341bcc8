// mf_host.cpp -- ORACLE / TEST INFRASTRUCTURE (not shipped, not linked into libmgbhip.so).
// A plain host sparse Cholesky that follows the plan produced by the product's symbolic
// analysis (csrc/mf_analysis.cpp: same scatter lists, relative indices, level order as the
// device kernels in csrc/mf_numeric.hip).  Two uses: (1) the symbolic analysis is validated
// on a machine without a GPU against SciPy's SuperLU; (2) it is the direct solver of the
// CPU baseline in bench.py, standing in for the CHOLMOD call behind the reference's
// `solve(symmetric(H), g)` (src/utils.jl:142-145) -- SuperLU alone would understate the
// reference's CPU path by an order of magnitude.
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../multigridbarrier.jl_amd/csrc/mf_analysis.hpp"
#include <map>
#include <memory>

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


// ---- cached-plan variant for the CPU baseline: analyze once per pattern, then factor+solve ----
struct HostSolver {
    MfPlan plan;
    std::vector<double> arena, uvec;
};

extern "C" void* mf_host_analyze(int64_t n, const int32_t* rowptr, const int32_t* colidx, int32_t leaf_size) {
    auto* S = new HostSolver();
    MfOptions opt;
    if (leaf_size > 0) opt.leaf_size = leaf_size;
    if (const char* e = getenv("MF_SEPW")) opt.sep_weight = atof(e);
    if (const char* e = getenv("MF_MERGE")) opt.merge_max_m = atoi(e);
    try {
        mf_analyze(n, rowptr, colidx, opt, S->plan);
    } catch (const std::exception&) {
        delete S;
        return nullptr;
    }
    S->arena.assign((size_t)S->plan.arena_doubles, 0.0);
    S->uvec.assign((size_t)S->plan.uvec_doubles, 0.0);
    return S;
}

extern "C" void mf_host_free(void* h) { delete (HostSolver*)h; }

extern "C" int mf_host_factor_solve(void* h, const double* values, const double* b, double* x) {
    HostSolver* S = (HostSolver*)h;
    const MfPlan& plan = S->plan;
    std::vector<double>& arena = S->arena;
    std::vector<double>& uvec = S->uvec;
    std::fill(arena.begin(), arena.end(), 0.0);
    int status = 0;
    const int32_t nf = (int32_t)plan.fronts.size();
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
            for (int32_t j = kc; j < mc; ++j) {
                double* Fc = F + (int64_t)rel[j - kc] * m;
                const double* Uc = U + (int64_t)j * mc;
                for (int32_t r = j; r < mc; ++r) Fc[rel[r - kc]] += Uc[r];
            }
        }
        for (int32_t j = 0; j < k; ++j) {
            double* Lj = F + (int64_t)j * m;
            double d = Lj[j];
            if (!(d > 0)) status = 3;
            double l = std::sqrt(d), inv = 1.0 / l;
            Lj[j] = l;
            for (int32_t r = j + 1; r < m; ++r) Lj[r] *= inv;
            for (int32_t c2 = j + 1; c2 < m; ++c2) {
                const double lc = Lj[c2];
                if (lc == 0.0) continue;
                double* Fc = F + (int64_t)c2 * m;
                for (int32_t r = c2; r < m; ++r) Fc[r] -= Lj[r] * lc;
            }
        }
    }
    const int64_t n = plan.n;
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
            const double* Lj = F + (int64_t)j * m;
            for (int32_t r = j + 1; r < m; ++r) t[r] -= Lj[r] * t[j];
        }
        for (int32_t j = 0; j < k; ++j) y[idx[j]] = t[j];
        for (int32_t j = k; j < m; ++j) uvec[f.u_off + j - k] = t[j];
    }
    for (int32_t i = nf - 1; i >= 0; --i) {
        const Front& f = plan.fronts[i];
        const double* F = arena.data() + f.F_off;
        const int32_t* idx = plan.front_idx.data() + f.idx_off;
        const int32_t m = f.m, k = f.k;
        t.assign(m, 0.0);
        for (int32_t j = 0; j < k; ++j) t[j] = y[idx[j]];
        for (int32_t j = k; j < m; ++j) t[j] = x[idx[j]];
        for (int32_t j = k - 1; j >= 0; --j) {
            const double* Lj = F + (int64_t)j * m;
            double s = t[j];
            for (int32_t r = j + 1; r < m; ++r) s -= Lj[r] * t[r];
            t[j] = s / Lj[j];
        }
        for (int32_t j = 0; j < k; ++j) x[idx[j]] = t[j];
    }
    return status;
}

// plan inspection for tests / tuning: per front (level, m, k, nchild)
extern "C" int64_t mf_host_plan_fronts(void* h, int32_t* out4, int64_t cap) {
    HostSolver* S = (HostSolver*)h;
    const int64_t nf = (int64_t)S->plan.fronts.size();
    for (int64_t i = 0; i < nf && i < cap; ++i) {
        const Front& f = S->plan.fronts[i];
        out4[4 * i + 0] = f.level; out4[4 * i + 1] = f.m; out4[4 * i + 2] = f.k; out4[4 * i + 3] = f.nchild;
    }
    return nf;
}


// Hash of every array of the plan the product's symbolic analysis produces for a pattern (tests: the threaded parts of
// the analysis must reproduce the serial plan exactly; MGBHIP_ANALYZE_THREADS selects the thread count per process).
extern "C" uint64_t mf_host_plan_hash(int64_t n, const int32_t* rowptr, const int32_t* colidx, int32_t border) {
    MfPlan plan;
    MfOptions opt;
    opt.border = border != 0;
    opt.protect_peeled = true;
    try {
        mf_analyze(n, rowptr, colidx, opt, plan);
    } catch (const std::exception&) {
        return 0;
    }
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const void* p, size_t bytes) {
        const unsigned char* c = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    };
    for (const Front& f : plan.fronts) {
        const int64_t v[] = {f.k, f.m, f.level, f.parent, f.nchild, f.a_cnt, f.F_off, f.idx_off, f.u_off, f.child_off, f.rel_off, f.a_off, f.acol_off};
        mix(v, sizeof(v));
    }
    mix(plan.front_idx.data(), plan.front_idx.size() * sizeof(int32_t));
    mix(plan.children.data(), plan.children.size() * sizeof(int32_t));
    mix(plan.rel.data(), plan.rel.size() * sizeof(int32_t));
    mix(plan.a_src.data(), plan.a_src.size() * sizeof(int32_t));
    mix(plan.a_dst.data(), plan.a_dst.size() * sizeof(int32_t));
    mix(plan.a_colptr.data(), plan.a_colptr.size() * sizeof(int32_t));
    mix(plan.level_ptr.data(), plan.level_ptr.size() * sizeof(plan.level_ptr[0]));
    return h ? h : 1;
}
