// driver.cpp -- the host control flow of the hot path: damped Newton with line search and
// stopping rules (reference: src/newton.jl:227-287, :139-154, :84-103, :187, :222-225), the
// level sweep mgb_step (src/mgb.jl:10-82), the t-ramp mgb_core (src/mgb.jl:91-183) and
// _matched_t (src/mgb.jl:307-330).  Vectors never leave the device; the host sees only the
// scalars the reference's control flow branches on (objective, decrement, norms, flags).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>

#include "problem.hpp"

using namespace mgbhip;

namespace {

constexpr double EPS = std::numeric_limits<double>::epsilon();

// MGBHIP_DEBUG=1 prints the scalars the Newton control flow branches on (the reference's
// @mgblog lines, src/newton.jl:256) to stderr.
bool debug_on() {
    static const bool on = [] { const char* e = getenv("MGBHIP_DEBUG"); return e && e[0] == '1'; }();
    return on;
}
#define DBG(...) do { if (debug_on()) { fprintf(stderr, __VA_ARGS__); } } while (0)

struct Stop {       // stopping_exact / stopping_inexact, or the caller's own rule
    double lambda_tol;   // < 0: exact only
    double theta;
    int (*fn)(double, double, double, double, double, double, void*) = nullptr;
    void* user = nullptr;
    bool operator()(double ymin, double ynext, double gmin, double gnorm_next, double ndecmin, double ndec) const {
        if (fn) return fn(ymin, ynext, gmin, gnorm_next, ndecmin, ndec, user) != 0;
        const bool exact = (ynext >= ymin) && (gnorm_next >= theta * gmin);
        if (lambda_tol >= 0) return (ndec < lambda_tol) || exact;
        return exact;
    }
};

struct NewtonResult {
    int k = 0;
    bool converged = false;
    double y = 0;
};

struct VecStats { double sumsq, bad; };

// One process per GPU (include/mgbhip.h, mgbhip_problem_set_collective): vectors live on this rank's unknowns with the
// interface replicated, so every reduction runs over the entries the rank owns (mask) and is then summed over ranks.
VecStats vec_stats(mgbhip_problem* P, int level, const double* d_v, int64_t len) {
    hipStream_t st = P->stream();
    launch_vec_stats(d_v, len, P->d_scratch.p, P->d_scal.p + 2, st, P->own_mask(level));
    P->read_scalars(2, 2);
    if (P->sharded()) P->allreduce_host(P->pin.d + 2, 2, 0);
    return VecStats{P->pin.d[2], P->pin.d[3]};
}

double dev_dot(mgbhip_problem* P, int level, const double* a, const double* b, int64_t len) {
    hipStream_t st = P->stream();
    launch_dot(a, b, len, P->d_scratch.p, P->d_scal.p + 4, st, P->own_mask(level));
    P->read_scalars(4, 1);
    if (P->sharded()) P->allreduce_host(P->pin.d + 4, 1, 0);
    return P->pin.d[4];
}

struct NewtonCtx {
    mgbhip_problem* P;
    int level;
    const double* d_zJ;   // snapshot of the fine iterate
    const double* d_c;
    int64_t m;
    double F0(const double* d_s) { return P->eval_f0(level, d_s, d_zJ, d_c); }
    // d_part: where a sharded problem keeps this rank's partial sums of the gradient (the right-hand side of its local elimination)
    void F1(const double* d_s, double* d_out, double* d_part) { P->eval_f1(level, d_s, d_zJ, d_c, d_out, P->sharded() ? d_part : nullptr); }
    const double* rhs_of_g() const { return P->sharded() ? P->d_gpart.p : P->d_g.p; }
    // H stays in the slab where the level allows; g (the right-hand side of the solve that follows) lets the fine level
    // condense the element-local unknowns inside the element kernel
    void F2(const double* d_s) { P->eval_f2(level, d_s, d_zJ, d_c, false, rhs_of_g()); }
};

// One line-search trial shared by both searches: evaluates F0/F1 at xn (already formed in
// P->d_xn), rejects non-finite values like the reference's `error(...)` + catch protocol.
// F0, F1 and the gradient statistics are launched back to back and read with ONE
// synchronisation (value, sum of squares, non-finite count, and the step kernel's "moved" flag):
// evaluating F1 at a point F0 rejects only produces NaN/Inf that nobody reads.
bool trial_values(NewtonCtx& C, double& ynext, double& gnorm_next, int32_t* moved = nullptr, const double* step = nullptr) {
    mgbhip_problem* P = C.P;
    hipStream_t st = P->stream();
    // step != nullptr: the trial point x - (*step) n has not been formed yet.  On a selection level the element kernel forms it
    // on the fly and the step kernel that materialises it (and raises the "moved" stamp) runs behind the evaluation: the first
    // launch after the host's decision is the long one, and the host has submitted the rest before it ends.
    const bool onfly = step && P->can_fuse_step(C.level);
    if (step) {
        ++P->step_stamp;
        if (!onfly) launch_step(P->d_x.p, P->d_nv.p, *step, P->d_xn.p, C.m, P->d_flag.p, P->step_stamp, st);
        P->touch();
        if (onfly) { P->trial_x = P->d_x.p; P->trial_dir = P->d_nv.p; P->trial_alpha = *step; }
    }
    // the element kernel leaves its workgroup partials of f0; their sum, |g|^2, the non-finite count and the step kernel's
    // "moved" stamp are finished by ONE launch that also stores them in the pinned block (no reduce / copy launches)
    const bool fused = !P->dense;
    static const bool fuse_restrict = [] { const char* e = getenv("MGBHIP_NO_FUSED_RESTRICT"); return !(e && e[0] == '1'); }();
    P->trial_fuse = mgbhip_problem::TrialFuse();
    if (fused && fuse_restrict) {           // restriction + |g|^2 partials (+ the deferred step) in one launch where the level allows it
        P->trial_fuse.want = true;
        if (onfly) {
            P->trial_fuse.x = P->d_x.p; P->trial_fuse.n = P->d_nv.p; P->trial_fuse.s = *step; P->trial_fuse.xn = P->d_xn.p;
            P->trial_fuse.moved = P->d_flag.p; P->trial_fuse.stamp = P->step_stamp;
        }
    }
    P->eval_f01_launch(C.level, P->d_xn.p, C.d_zJ, C.d_c, P->d_gn.p, P->sharded() ? P->d_gnpart.p : nullptr, fused);
    const bool restrict_fused = P->trial_fuse.done;
    P->trial_fuse.want = false;
    if (onfly) {
        P->trial_x = nullptr; P->trial_dir = nullptr;
        if (!restrict_fused) launch_step(P->d_x.p, P->d_nv.p, *step, P->d_xn.p, C.m, P->d_flag.p, P->step_stamp, st);
    }
    const double seq = P->next_seq();
    launch_trial_finish(P->d_gn.p, C.m, P->d_scratch.p, fused ? P->d_partials.p : nullptr, elem_grid(P->p, P->N), P->d_scal.p,
                        P->d_flag.p, P->pin.dev, st, P->own_mask(C.level), seq, restrict_fused);
    P->wait_results(seq);
    P->pin.i[0] = P->pin.d[4] == (double)P->step_stamp ? 1 : 0;      // the stamp of the step kernel that formed this trial point
    if (P->sharded()) {              // value, |g|^2, non-finite count and the "moved" flag in one sum over ranks
        P->pin.d[1] = (double)P->pin.i[0];
        P->allreduce_host(P->pin.d, 4, 0);
        P->pin.i[0] = P->pin.d[1] != 0.0 ? 1 : 0;
    }
    if (moved) *moved = P->pin.i[0];
    ynext = P->pin.d[0];
    if (!std::isfinite(ynext)) return false;
    if (P->pin.d[3] != 0.0 || !std::isfinite(P->pin.d[2])) return false;
    gnorm_next = std::sqrt(P->pin.d[2]);
    return true;
}

// linesearch_backtracking (src/newton.jl:139-154 with _linesearch_loop :35-50).
// On return P->d_xn / d_gn hold the accepted (or last computed) trial; returns false when no
// trial was ever evaluated successfully (then x, y, g stay as they were).
bool linesearch_backtracking(NewtonCtx& C, const mgbhip_options& opt, double y, double inc, double& ynext,
                             double& gnorm_next) {
    mgbhip_problem* P = C.P;
    hipStream_t st = P->stream();
    double s = 1.0;
    bool have = false;
    while (s > 0.0) {
        double yn, gn;
        int32_t moved = 0;
        if (trial_values(C, yn, gn, &moved, &s)) {
            have = true;
            ynext = yn;
            gnorm_next = gn;
            const bool stalled = (moved == 0);
            if (stalled || yn <= y - opt.ls_c1 * inc * s) return true;
        }
        s *= opt.ls_beta;
    }
    return have;
}

// linesearch_illinois (src/newton.jl:84-103, illinois :4-27)
bool linesearch_illinois(NewtonCtx& C, const mgbhip_options& opt, double inc, double& ynext, double& gnorm_next) {
    mgbhip_problem* P = C.P;
    hipStream_t st = P->stream();
    struct Reject {};
    auto phi = [&](double sigma) -> double {
        launch_step(P->d_x.p, P->d_nv.p, sigma, P->d_xn.p, C.m, P->d_flag.p, ++P->step_stamp, st);
        P->touch();
        const double f = C.F0(P->d_xn.p);
        if (!std::isfinite(f)) throw Reject();
        C.F1(P->d_xn.p, P->d_gn.p, P->d_gnpart.p);
        const double v = dev_dot(P, C.level, P->d_gn.p, P->d_nv.p, C.m);
        if (!std::isfinite(v)) throw Reject();
        return v;
    };
    double s = 1.0;
    bool have = false;
    while (s > 0.0) {
        try {
            // illinois(phi, 0, s; fa = inc)
            double a = 0.0, b = s, fa = inc, fb = phi(s), root = b;
            bool done = false;
            if (fa == 0) { root = a; done = true; }
            else if (fa * fb >= 0) { root = b; done = true; }
            for (int it = 0; it < 10000 && !done; ++it) {
                const double c = (a * fb - b * fa) / (fb - fa);
                const double fc = phi(c);
                if (c <= std::fmin(a, b) || c >= std::fmax(a, b) || fc * fa == 0 || fc * fb == 0) { root = c; done = true; break; }
                if (fb * fc < 0) { a = b; fa = fb; } else { fa /= 2; }
                b = c; fb = fc;
            }
            if (!done) throw Reject();
            launch_step(P->d_x.p, P->d_nv.p, root, P->d_xn.p, C.m, P->d_flag.p, ++P->step_stamp, st);
        P->touch();
            double yn, gn;
            if (!trial_values(C, yn, gn)) throw Reject();
            ynext = yn;
            gnorm_next = gn;
            return true;
        } catch (Reject&) {
        }
        s *= opt.ls_beta;
    }
    return have;
}

NewtonResult newton(NewtonCtx& C, const mgbhip_options& opt, const Stop& stop, int maxit) {
    mgbhip_problem* P = C.P;
    hipStream_t st = P->stream();
    NewtonResult R;
    P->d_x.zero(st, (size_t)C.m);                        // s0 = zeros (src/mgb.jl:45)
    P->touch();                                          // new level / new fine snapshot: drop the cached z0 + R*s
    double y = C.F0(P->d_x.p);
    if (!std::isfinite(y)) throw InvalidArgument("newton: initial objective value is not finite");
    double ymin = y;
    if (P->sharded()) { P->d_gpart.ensure((size_t)C.m); P->d_gnpart.ensure((size_t)C.m); }
    C.F1(P->d_x.p, P->d_g.p, P->d_gpart.p);
    VecStats gs = vec_stats(P, C.level, P->d_g.p, C.m);
    if (gs.bad != 0.0 || !std::isfinite(gs.sumsq)) throw InvalidArgument("newton: initial gradient has non-finite entries");
    double gnorm = std::sqrt(gs.sumsq);
    double gmin = gnorm;
    double incmin = INFINITY;
    int k = 0;
    bool converged = false;
    Level& L = P->levels[C.level];
    struct RobustReset {          // the next Newton solve starts on the fast kernels again, whichever way this one ends
        MfSolver& s;
        ~RobustReset() { s.robust = false; }
    } robust_reset{L.solver};
    while (k < maxit && !converged) {
        ++k;
        C.F2(P->d_x.p);
        auto t0 = std::chrono::steady_clock::now();
        int fstatus = MGBHIP_OK;
        int leaf_flag = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            P->factor(C.level, C.rhs_of_g());               // the gradient rides along: no forward sweep afterwards
            P->trisolve_carried(C.level, P->d_nv.p);
            // pivot flags, direction statistics and lambda^2 = <g, n> in one round trip: the finishing launch stores them in
            // the pinned block and clears the solver's flags for the next factorization (no copy / memset launches)
            const double seq = P->next_seq();
            launch_dir_finish(P->d_nv.p, P->d_g.p, C.m, P->d_scratch.p, P->d_scal.p, L.solver.status_flags_rw(), P->pin.dev, st,
                              P->own_mask(C.level), seq);
            L.solver.flags_cleared();
            P->wait_results(seq);
            P->pin.i[1] = P->pin.d[5] != 0.0 ? 1 : 0;
            leaf_flag |= P->pin.d[6] != 0.0 ? 1 : 0;        // the condensed leaves are not re-formed by a refactorization: their flag sticks
            P->pin.i[2] = leaf_flag;
            fstatus = MfSolver::status_from(P->pin.i + 1, L.solver.factored_condensed);
            if (P->sharded()) {          // every rank must take the same branch: the pivot flag travels with the sums
                P->pin.d[5] = fstatus != MGBHIP_OK ? 1.0 : 0.0;
                P->allreduce_host(P->pin.d + 2, 4, 0);
                fstatus = P->pin.d[5] != 0.0 ? MGBHIP_ERR_NOT_SPD : MGBHIP_OK;
            }
            // The fast large-front kernels apply inverted 32 x 32 diagonal blocks; near the edge of singularity
            // that loses digits a substitution keeps.  A failed pivot, a non-finite direction or lambda^2 <= 0
            // is re-done once with the substitution kernels before the reference's own tests see it.
            // (lambda^2 <= 0 within EPS * max(|y|, 1) is the reference's legitimate round-off-floor exit, src/newton.jl:257-271)
            const bool at_floor = std::fabs(P->pin.d[4]) <= EPS * std::fmax(std::fabs(y), 1.0);
            const bool suspicious = fstatus != MGBHIP_OK || P->pin.d[3] != 0.0 || !std::isfinite(P->pin.d[2]) ||
                                    (!(P->pin.d[4] > 0) && !at_floor);
            if (!suspicious || attempt == 1 || !L.solver.has_inverse_path() || L.solver.robust) break;
            DBG("newton[lev %d] k=%d: direction rejected (status %d, lambda^2=%.3e): refactoring with the substitution kernels\n",
                C.level, k, fstatus, P->pin.d[4]);
            L.solver.robust = true;
            L.factored = false;
        }
        if (fstatus != MGBHIP_OK && P->lu_fallback(C.level, P->d_g.p, P->d_nv.p)) {
            // An exactly zero / non-finite pivot ended the un-pivoted LDL' (twice: inverse-based and substitution kernels).
            // Julia's `Symmetric(H) \ g` falls through Cholesky -> LDL' -> LU (src/utils.jl:142-145): so does this solve, with a
            // dense partially pivoted LU on the device, for systems small enough to be held densely.  The reference's own
            // tests (non-finite direction, lambda^2 <= 0) then see that direction.
            DBG("newton[lev %d] k=%d: LDL' met a zero pivot; direction from the pivoted LU fallback\n", C.level, k);
            launch_dir_finish(P->d_nv.p, P->d_g.p, C.m, P->d_scratch.p, P->d_scal.p, L.solver.status_flags_rw(), P->pin.dev, st,
                              P->own_mask(C.level));
            L.solver.flags_cleared();
            MGB_HIP_CHECK(hipStreamSynchronize(st));
            fstatus = MGBHIP_OK;
        }
        P->cnt.solve_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (fstatus != MGBHIP_OK) {
            DBG("newton[lev %d] k=%d: Cholesky met a non-positive pivot (y=%.17g |g|=%.6e)\n", C.level, k, y, gnorm);
            // H numerically not SPD: the reference's `\` would fall back to LDLt/LU and then
            // either report lambda^2 <= 0 or throw; both end this Newton attempt unconverged.
            converged = false;
            break;
        }
        if (P->pin.d[3] != 0.0 || !std::isfinite(P->pin.d[2])) throw InvalidArgument("newton: Newton direction has non-finite entries");
        const double inc = P->pin.d[4];
        DBG("newton[lev %d] k=%d y=%.17g |g|=%.6e lambda^2=%.6e\n", C.level, k, y, gnorm, inc);
        if (inc <= 0) {
            converged = std::fabs(inc) <= EPS * std::fmax(std::fabs(y), 1.0);   // src/newton.jl:257-271
            break;
        }
        double ynext = y, gnorm_next = gnorm;
        bool moved = (opt.line_search == 1) ? linesearch_illinois(C, opt, inc, ynext, gnorm_next)
                                            : linesearch_backtracking(C, opt, y, inc, ynext, gnorm_next);
        if (!moved) {          // every trial rejected: (xnext, ynext, gnext) = (x, y, g)
            ynext = y;
            gnorm_next = gnorm;
        }
        if (stop(ymin, ynext, gmin, gnorm_next, std::sqrt(incmin), std::sqrt(inc))) converged = true;
        if (moved) {
            std::swap(P->d_x, P->d_xn);
            std::swap(P->d_g, P->d_gn);
            std::swap(P->d_gpart, P->d_gnpart);
        }
        y = ynext;
        gnorm = gnorm_next;
        gmin = std::fmin(gmin, gnorm);
        ymin = std::fmin(ymin, y);
        incmin = std::fmin(inc, incmin);
    }
    DBG("newton[lev %d] done k=%d converged=%d y=%.17g\n", C.level, k, (int)converged, y);
    R.k = k;
    R.converged = converged;
    R.y = y;
    P->cnt.newton += k;
    return R;
}

struct StepResult {
    bool converged;
    std::vector<int64_t> its;
};

// mgb_step (src/mgb.jl:16-82); P->d_z is updated in place on converged level solves.
StepResult mgb_step(mgbhip_problem* P, const double* d_c, const mgbhip_options& opt, bool finalize_now,
                    bool initial_step) {
    const int L = (int)P->levels.size();
    StepResult out;
    out.its.assign(L, 0);
    hipStream_t st = P->stream();
    const Stop sc{opt.stop_lambda_tol, opt.stop_theta, opt.stopping_criterion, opt.user};
    const Stop fin{-1.0, opt.finalize_theta};
    const size_t zn = (size_t)P->nu * P->n;
    auto eta = [&](int j, int J, const Stop& stop, int maxit) -> bool {
        (void)j;
        const int lev = J - 1;
        // snapshot zJ = z (the Newton closures capture it, src/mgb.jl:48)
        MGB_HIP_CHECK(hipMemcpyAsync(P->d_z0.p, P->d_z.p, zn * sizeof(double), hipMemcpyDeviceToDevice, st));
        NewtonCtx C{P, lev, P->d_z0.p, d_c, P->levels[lev].m};
        NewtonResult r = newton(C, opt, stop, maxit);
        out.its[lev] += r.k;
        if (r.converged) {
            const Level& Lv = P->levels[lev];
            StageScope scp(P->ctx->timers, "prolong");
            launch_csr_matvec(Lv.rows, Lv.Rptr.p, Lv.Rcol.p, Lv.Rval.p, P->d_x.p, P->d_z.p, true, false, st);
        }
        return r.converged;
    };
    std::function<bool(int, int)> dac = [&](int j, int J) -> bool {   // divide_and_conquer (src/mgb.jl:10-15)
        const int mn = (initial_step && J - j == 1) ? opt.maxit : opt.max_newton;
        if (eta(j, J, sc, mn)) return true;
        const int jmid = (j + J) / 2;
        if (jmid == j || jmid == J) return false;
        return dac(j, jmid) && dac(jmid, J);
    };
    bool converged = dac(0, L);
    if (finalize_now) {
        const bool foo = eta(L - 1, L, fin, opt.maxit);
        converged = converged && foo;
    }
    out.converged = converged;
    return out;
}

double c_dot_Dz(mgbhip_problem* P, const double* d_c) {
    // sum_j dot(w .* c[:, j], (D_j z))  (src/mgb.jl:135-136, :165-166) with the UNSCALED cost grid
    // c (not t*c): the linear part of f0 at s = 0; invn = 0 switches the barrier term off
    hipStream_t st = P->stream();
    ElemParams E = P->base_params(-1, nullptr, P->d_z.p, d_c);
    E.invn = 0.0;
    E.bw = nullptr;
    launch_elem(E, MODE_F0, st);
    launch_reduce_partials(P->d_partials.p, elem_grid(P->p, P->N), P->d_scal.p, st);
    double v;
    P->d_scal.download(&v, 1, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    if (P->sharded()) P->allreduce_host(&v, 1, 0);
    return v;
}

bool slack_feasible(mgbhip_problem* P, std::vector<double>& hbuf) {
    // feasible(z) = maximum(WW*z) < 0 with WW selecting the last (slack) component (src/mgb.jl:454)
    hipStream_t st = P->stream();
    hbuf.resize((size_t)P->n);
    MGB_HIP_CHECK(hipMemcpyAsync(hbuf.data(), P->d_z.p + (size_t)(P->nu - 1) * P->n, (size_t)P->n * sizeof(double),
                                 hipMemcpyDeviceToHost, st));
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    double mx = -INFINITY;
    for (double v : hbuf) mx = std::fmax(mx, v);
    if (P->sharded()) P->allreduce_host(&mx, 1, 1);
    return mx < 0;
}

}  // namespace

int core_run(mgbhip_problem* P, double* z, const double* c, const mgbhip_options* optp, mgbhip_core_result* res) {
    const mgbhip_options opt = *optp;
    hipStream_t st = P->stream();
    const size_t zn = (size_t)P->nu * P->n;
    const size_t cn = (size_t)P->n * P->nD;
    const int L = (int)P->levels.size();
    P->prepare_all();
    P->d_z.upload(z, zn, st);
    P->d_c0.upload(c, cn, st);
    P->cnt = Counters();
    auto tb = std::chrono::steady_clock::now();
    auto now = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count(); };
    double t = opt.t;
    const double target = 1.0 / opt.tol;
    double kappa = opt.kappa;
    const double kappa0 = kappa;
    int k = 1;
    std::vector<double> hbuf;
    double t_first = INFINITY;
    std::vector<double> zhost;
    auto early = [&](double tt) -> bool {
        if (opt.early_stop_fn) {                           // user `early_stop(z)` / `early_stop(z, t)` (src/mgb.jl:85-89)
            zhost.resize(zn);
            P->d_z.download(zhost.data(), zn, st);
            MGB_HIP_CHECK(hipStreamSynchronize(st));
            if (opt.early_stop_fn(zhost.data(), tt, opt.user) != 0) return true;
        }
        if (opt.early_stop != 1) return false;
        if (!slack_feasible(P, hbuf)) return false;        // margin rule (src/mgb.jl:486-491)
        t_first = std::fmin(t_first, tt);
        return tt >= 2 * t_first;
    };
    auto record = [&](int kk, const std::vector<int64_t>& its, bool add) {
        if (kk - 1 < res->cap_steps && res->its) {
            for (int l = 0; l < L; ++l) {
                if (add) res->its[(size_t)(kk - 1) * L + l] += its[l];
                else res->its[(size_t)(kk - 1) * L + l] = its[l];
            }
        }
    };
    auto setc = [&](double tt) { launch_scale_copy(P->d_c0.p, tt, P->d_c.p, (int64_t)cn, st); };
    res->failure_code = 0;
    res->L = L;
    if (res->times && res->cap_steps > 0) res->times[0] = now();
    setc(t);
    StepResult S = mgb_step(P, P->d_c.p, opt, opt.finalize && t >= target, true);
    if (!S.converged) {
        res->failure_code = 1;   // :stall -- "Initial centering failed"
        res->k = 1;
        res->t_final = t;
        res->t_elapsed = now();
        return MGBHIP_ERR_CONVERGENCE;
    }
    record(1, S.its, false);
    auto diag = [&](int kk) {
        if (kk - 1 < res->cap_steps) {
            if (res->ts) res->ts[kk - 1] = t;
            if (res->kappas) res->kappas[kk - 1] = kappa;
            if (res->c_dot_Dz) res->c_dot_Dz[kk - 1] = c_dot_Dz(P, P->d_c0.p);
        }
    };
    diag(1);
    bool stopped_early = false;
    while (t < target && kappa > 1 && k < opt.maxit) {
        if (early(t)) { stopped_early = true; break; }
        ++k;
        if (k - 1 < res->cap_steps && res->its)
            for (int l = 0; l < L; ++l) res->its[(size_t)(k - 1) * L + l] = 0;
        if (res->times && k - 1 < res->cap_steps) res->times[k - 1] = now();
        while (kappa > 1) {
            const double t1 = kappa * t;
            setc(t1);
            // a failed step must not move z: keep a copy to roll back to
            P->d_tmp.ensure(zn);
            MGB_HIP_CHECK(hipMemcpyAsync(P->d_tmp.p, P->d_z.p, zn * sizeof(double), hipMemcpyDeviceToDevice, st));
            S = mgb_step(P, P->d_c.p, opt, opt.finalize && t1 >= target, false);
            record(k, S.its, true);
            if (S.converged) {
                int64_t mx = 0;
                for (int64_t v : S.its) mx = std::max(mx, v);
                if ((double)mx <= opt.max_newton * 0.5) kappa = std::fmin(kappa0, kappa * kappa);
                t = t1;
                break;
            }
            // z = SOL.z only on success (src/mgb.jl:150-157): restore
            MGB_HIP_CHECK(hipMemcpyAsync(P->d_z.p, P->d_tmp.p, zn * sizeof(double), hipMemcpyDeviceToDevice, st));
            kappa = std::sqrt(kappa);
        }
        diag(k);
    }
    bool converged = (t >= target) || stopped_early || early(t);
    res->k = k;
    res->t_final = t;
    res->t_elapsed = now();
    res->solve_seconds = P->cnt.solve_seconds;
    res->newton_iterations = P->cnt.newton;
    res->f0_evals = P->cnt.f0;
    res->f1_evals = P->cnt.f1;
    res->f2_evals = P->cnt.f2;
    res->factorizations = P->cnt.factor;
    P->d_z.download(z, zn, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    if (!converged) {
        res->failure_code = (kappa <= 1) ? 1 : 2;
        return MGBHIP_ERR_CONVERGENCE;
    }
    return MGBHIP_OK;
}

int matched_t_run(mgbhip_problem* P, const double* z, const double* c, double t_default, double* t_out) {
    // _matched_t (src/mgb.jl:307-330): two Hessian solves at the fine level, s = 0
    hipStream_t st = P->stream();
    const size_t zn = (size_t)P->nu * P->n;
    const size_t cn = (size_t)P->n * P->nD;
    const int lev = (int)P->levels.size() - 1;
    const int64_t m = P->levels[lev].m;
    P->d_z0.upload(z, zn, st);
    P->d_c0.upload(c, cn, st);
    P->d_c.zero(st, cn);
    P->d_x.zero(st, (size_t)m);
    P->touch();
    *t_out = t_default;
    // gphi = f1(c = 0); gc = f1(c) - gphi; H = f2
    const bool sh = P->sharded();
    if (sh) { P->d_gpart.ensure((size_t)m); P->d_gnpart.ensure((size_t)m); }
    P->eval_f1(lev, P->d_x.p, P->d_z0.p, P->d_c.p, P->d_g.p, sh ? P->d_gpart.p : nullptr);      // gphi -> d_g
    P->eval_f1(lev, P->d_x.p, P->d_z0.p, P->d_c0.p, P->d_gn.p, sh ? P->d_gnpart.p : nullptr);   // f1(c) -> d_gn
    launch_axpy(-1.0, P->d_g.p, P->d_gn.p, m, st);                      // gc -> d_gn
    if (sh) launch_axpy(-1.0, P->d_gpart.p, P->d_gnpart.p, m, st);
    P->eval_f2(lev, P->d_x.p, P->d_z0.p, P->d_c0.p);
    if (sh) {
        // domain decomposition: only the bordered factorization crosses ranks (the interface front carries the
        // right-hand side), so each of the two solves is a factorization with its own border
        P->d_tmp.ensure((size_t)m + 1);
        P->factor(lev, P->d_gpart.p);
        P->trisolve_carried(lev, P->d_tmp.p);
        int bad = P->levels[lev].solver.status(st) != MGBHIP_OK;
        MGB_HIP_CHECK(hipMemcpyAsync(P->d_nv.p, P->d_tmp.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, st));
        P->levels[lev].factored = false;
        P->factor(lev, P->d_gnpart.p);
        P->trisolve_carried(lev, P->d_tmp.p);
        bad |= P->levels[lev].solver.status(st) != MGBHIP_OK;
        MGB_HIP_CHECK(hipMemcpyAsync(P->d_xn.p, P->d_tmp.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, st));
        double fb = bad ? 1.0 : 0.0;
        P->allreduce_host(&fb, 1, 0);
        if (fb != 0.0) return MGBHIP_OK;
    } else {
        P->factor(lev);
        if (P->levels[lev].solver.status(st) != MGBHIP_OK) return MGBHIP_OK;   // degenerate: keep t_default
        P->trisolve(lev, P->d_g.p, P->d_nv.p);                              // nphi
        P->trisolve(lev, P->d_gn.p, P->d_xn.p);                             // nc
    }
    const double d = dev_dot(P, lev, P->d_gn.p, P->d_xn.p, m);
    const double b = dev_dot(P, lev, P->d_g.p, P->d_xn.p, m) + dev_dot(P, lev, P->d_gn.p, P->d_nv.p, m);
    if (!(d > 0)) return MGBHIP_OK;
    const double tstar = -b / (2 * d);
    if (!(std::isfinite(tstar) && tstar > 0)) return MGBHIP_OK;
    *t_out = std::fmin(std::fmax(tstar, std::sqrt(EPS)), t_default);
    return MGBHIP_OK;
}
