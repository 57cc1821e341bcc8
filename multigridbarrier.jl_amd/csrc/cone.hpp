// cone.hpp -- per-node barrier functors on the device.
//
// The reference compiles arbitrary Julia functors into its map_rows kernel
// (ext/MultiGridBarrierCUDAExt/map_rows_gpu.jl:20-28); a C-ABI backend enumerates the
// functor families instead: Euclidean-power cones
// (src/convex_euclidian_power.jl:71-253, :387-433), linear inequalities
// (src/convex_linear.jl:119-214), their piecewise sums (src/convex_piecewise.jl:15-75)
// and the phase-I wrapper (src/mgb.jl:217-287).  Everything is evaluated in registers:
// index scatters are written as compile-time-unrolled selects so that no array is
// dynamically indexed (no scratch memory).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mgbhip.h"

namespace mgbhip {

struct PieceDev {
    int32_t kind, ni, nc;
    int32_t idx[MGBHIP_MAX_IDX];
    const double* A;
    const double* b;
    const double* p;
    const double* mu;
    const double* select;
    double p_const, mu_const;
};

struct ConeDev {
    int32_t npieces;
    int32_t feasibility;
    int32_t NC;
    double box_b, box_R;
    PieceDev pc[MGBHIP_MAX_PIECES];
};

// "Convex programmer's log" (src/utils.jl:14): -Inf off the domain instead of a throw.
__device__ __forceinline__ double mgb_Log(double x) { return x <= 0.0 ? -INFINITY : log(x); }
// src/convex_linear.jl:388-390
__device__ __forceinline__ double mgb_safe_pow(double s, double a) { return exp(a * mgb_Log(s)); }

// select y[i] for a runtime i without dynamic register indexing
template <int NY>
__device__ __forceinline__ double pick(const double (&y)[NY], int i) {
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < NY; ++j) v = (i == j) ? y[j] : v;
    return v;
}

// Accumulate one piece's barrier (or cobarrier when slack_pos >= 0: y[slack_pos] is the
// slack, added to s for EP and to every row for LINEAR) into F / g / H.
// ORDER: 0 value, 1 gradient, 2 Hessian.  H is row-major NY x NY (symmetric anyway).
template <int NY, int ORDER>
__device__ __forceinline__ void piece_accumulate(const PieceDev& P, int64_t node, int64_t n,
                                                 const double (&y)[NY], int slack_pos, double& F,
                                                 double (&g)[NY], double (&H)[NY * NY]) {
    constexpr int MI = MGBHIP_MAX_IDX;
    const int ni = P.ni;
    double yk[MI];
#pragma unroll
    for (int c = 0; c < MI; ++c) yk[c] = (c < ni) ? pick<NY>(y, P.idx[c]) : 0.0;
    const double slack = (slack_pos >= 0) ? pick<NY>(y, slack_pos) : 0.0;

    // local gradient / Hessian in the idx coordinates (+ slack row/col)
    double gl[MI], Hl[MI * MI], cross[MI];
    double g_sl = 0.0, H_sl = 0.0;
#pragma unroll
    for (int i = 0; i < MI; ++i) { gl[i] = 0.0; cross[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < MI * MI; ++i) Hl[i] = 0.0;

    if (P.kind == MGBHIP_KIND_EP) {
        const int nz = ni;
        // z = A yk + b ; A column-major nz x nz per node (NULL = identity)
        double z[MI];
        double Am[MI * MI];
        const bool hasA = P.A != nullptr;
#pragma unroll
        for (int r = 0; r < MI; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < MI; ++c) {
                double a = 0.0;
                if (r < nz && c < nz) a = hasA ? P.A[node + n * (int64_t)(r + nz * c)] : (r == c ? 1.0 : 0.0);
                Am[r + MI * c] = a;
                acc += a * yk[c];
            }
            z[r] = acc + ((P.b != nullptr && r < nz) ? P.b[node + n * (int64_t)r] : 0.0);
        }
        const double p0 = P.p ? P.p[node] : P.p_const;
        const double mu = P.mu ? P.mu[node] : P.mu_const;
        double qsq = 0.0;
#pragma unroll
        for (int r = 0; r < MI; ++r) qsq += (r < nz - 1) ? z[r] * z[r] : 0.0;
        const double s = pick<MI>(z, nz - 1) + slack;
        const double alpha = 2.0 / p0;
        const double s_a = mgb_safe_pow(s, alpha);
        const double rr = s_a - qsq;
        if (ORDER == 0) {
            F += -mgb_Log(rr) - mu * mgb_Log(s);
            return;
        }
        const double inv_r = 1.0 / rr;
        const double s_am1 = mgb_safe_pow(s, alpha - 1.0);
        double gz[MI], Hz[MI * MI];
        if (ORDER == 1) {
#pragma unroll
            for (int r = 0; r < MI; ++r)
                gz[r] = (r < nz - 1) ? 2.0 * inv_r * z[r] : ((r == nz - 1) ? (-alpha * s_am1 * inv_r - mu / s) : 0.0);
            // g_idx = A' gz
#pragma unroll
            for (int c = 0; c < MI; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < MI; ++r) acc += Am[r + MI * c] * gz[r];
                gl[c] = acc;
            }
            g_sl = pick<MI>(gz, nz - 1);
        } else {
            const double inv_r2 = inv_r * inv_r;
            const double coef_qs = -2.0 * alpha * s_am1 * inv_r2;
            const double s_am2 = mgb_safe_pow(s, alpha - 2.0);
            const double s_2am2 = mgb_safe_pow(s, 2.0 * alpha - 2.0);
            const double H_ss = -alpha * (alpha - 1.0) * s_am2 * inv_r + alpha * alpha * s_2am2 * inv_r2 + mu / (s * s);
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    double v = 0.0;
                    const bool iq = i < nz - 1, jq = j < nz - 1;
                    const bool is = i == nz - 1, js = j == nz - 1;
                    if (iq && jq) v = 4.0 * z[i] * z[j] * inv_r2 + (i == j ? 2.0 * inv_r : 0.0);
                    else if (iq && js) v = coef_qs * z[i];
                    else if (is && jq) v = coef_qs * z[j];
                    else if (is && js) v = H_ss;
                    Hz[i + MI * j] = v;
                }
            // Hl = A' Hz A ; cross = A' Hz[:, nz-1] ; H_sl = Hz[nz-1, nz-1]
            double T[MI * MI];   // T = Hz A
#pragma unroll
            for (int c = 0; c < MI; ++c)
#pragma unroll
                for (int r = 0; r < MI; ++r) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < MI; ++q) acc += Hz[r + MI * q] * Am[q + MI * c];
                    T[r + MI * c] = acc;
                }
#pragma unroll
            for (int c = 0; c < MI; ++c)
#pragma unroll
                for (int r = 0; r < MI; ++r) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < MI; ++q) acc += Am[q + MI * r] * T[q + MI * c];
                    Hl[r + MI * c] = acc;
                }
            double hcol[MI];
#pragma unroll
            for (int r = 0; r < MI; ++r) {
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < MI; ++q) v = (q == nz - 1) ? Hz[r + MI * q] : v;
                hcol[r] = v;
            }
#pragma unroll
            for (int c = 0; c < MI; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < MI; ++r) acc += Am[r + MI * c] * hcol[r];
                cross[c] = acc;
            }
            H_sl = H_ss;
        }
    } else {
        // linear inequalities: Fv = A yk + b (+ slack); A column-major nc x ni per node
        const int nc = P.nc;
        constexpr int MC = MGBHIP_MAX_IDX;
        double Am[MC * MI], Fv[MC];
        const bool hasA = P.A != nullptr;
#pragma unroll
        for (int r = 0; r < MC; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < MI; ++c) {
                double a = 0.0;
                if (r < nc && c < ni) a = hasA ? P.A[node + n * (int64_t)(r + nc * c)] : (r == c ? 1.0 : 0.0);
                Am[r + MC * c] = a;
                acc += a * yk[c];
            }
            Fv[r] = acc + ((P.b != nullptr && r < nc) ? P.b[node + n * (int64_t)r] : 0.0) + slack;
        }
        if (ORDER == 0) {
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < MC; ++r) acc += (r < nc) ? mgb_Log(Fv[r]) : 0.0;
            F += -acc;
            return;
        }
        if (ORDER == 1) {
            double invF[MC];
#pragma unroll
            for (int r = 0; r < MC; ++r) invF[r] = (r < nc) ? 1.0 / Fv[r] : 0.0;
#pragma unroll
            for (int c = 0; c < MI; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < MC; ++r) acc += Am[r + MC * c] * invF[r];
                gl[c] = -acc;
            }
#pragma unroll
            for (int r = 0; r < MC; ++r) g_sl -= invF[r];
        } else {
            double invF2[MC];
#pragma unroll
            for (int r = 0; r < MC; ++r) invF2[r] = (r < nc) ? 1.0 / (Fv[r] * Fv[r]) : 0.0;
#pragma unroll
            for (int c = 0; c < MI; ++c)
#pragma unroll
                for (int a = 0; a < MI; ++a) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < MC; ++r) acc += Am[r + MC * a] * invF2[r] * Am[r + MC * c];
                    Hl[a + MI * c] = acc;
                }
#pragma unroll
            for (int c = 0; c < MI; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < MC; ++r) acc += Am[r + MC * c] * invF2[r];
                cross[c] = acc;
            }
#pragma unroll
            for (int r = 0; r < MC; ++r) H_sl += invF2[r];
        }
    }

    // scatter into the y layout (src/convex_linear.jl:237-366)
    if (ORDER == 1) {
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            double add = 0.0;
#pragma unroll
            for (int c = 0; c < MI; ++c) add = (c < ni && P.idx[c] == i) ? gl[c] : add;
            if (i == slack_pos) add = g_sl;
            g[i] += add;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            int ki = -1;
#pragma unroll
            for (int c = 0; c < MI; ++c) ki = (c < ni && P.idx[c] == i) ? c : ki;
#pragma unroll
            for (int j = 0; j < NY; ++j) {
                int kj = -1;
#pragma unroll
                for (int c = 0; c < MI; ++c) kj = (c < ni && P.idx[c] == j) ? c : kj;
                double add = 0.0;
                if (i == slack_pos && j == slack_pos) add = H_sl;
                else if (i == slack_pos && kj >= 0) add = pick<MI>(cross, kj);
                else if (j == slack_pos && ki >= 0) add = pick<MI>(cross, ki);
                else if (ki >= 0 && kj >= 0) {
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q < MI * MI; ++q) v = (q == ki + MI * kj) ? Hl[q] : v;
                    add = v;
                }
                H[i * NY + j] += add;
            }
        }
    }
}

// Fast path of the most common Convex: a single Euclidean-power piece with A = I, b = 0, no
// select grid (the reference's default p-Laplace cone, src/mgb.jl:722).  Same formulas as
// core_grad / core_hess (src/convex_euclidian_power.jl:387-433) with the four powers of s
// derived from one exp/log pair: s^(a-1) = s^a / s, s^(a-2) = s^a / s^2, s^(2a-2) = (s^a / s)^2.
template <int NY, int ORDER>
__device__ __forceinline__ void ep_identity_eval(const PieceDev& P, int64_t node, const double (&y)[NY],
                                                 double& F, double (&g)[NY], double (&H)[NY * NY]) {
    constexpr int MI = MGBHIP_MAX_IDX;
    const int nz = P.ni;
    double z[MI];
#pragma unroll
    for (int c = 0; c < MI; ++c) z[c] = (c < nz) ? pick<NY>(y, P.idx[c]) : 0.0;
    const double p0 = P.p ? P.p[node] : P.p_const;
    const double mu = P.mu ? P.mu[node] : P.mu_const;
    double qsq = 0.0;
#pragma unroll
    for (int c = 0; c < MI; ++c) qsq += (c < nz - 1) ? z[c] * z[c] : 0.0;
    const double s = pick<MI>(z, nz - 1);
    const double alpha = 2.0 / p0;
    const double ls = mgb_Log(s);
    const double s_a = exp(alpha * ls);
    const double rr = s_a - qsq;
    if (ORDER == 0) {
        F = -mgb_Log(rr) - mu * ls;
        return;
    }
    const double inv_r = 1.0 / rr;
    const double inv_s = 1.0 / s;
    const double s_am1 = s_a * inv_s;
    double gz[MI], Hz[MI * MI];
    if (ORDER == 1) {
#pragma unroll
        for (int c = 0; c < MI; ++c)
            gz[c] = (c < nz - 1) ? 2.0 * inv_r * z[c] : ((c == nz - 1) ? (-alpha * s_am1 * inv_r - mu * inv_s) : 0.0);
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            double add = 0.0;
#pragma unroll
            for (int c = 0; c < MI; ++c) add = (c < nz && P.idx[c] == i) ? gz[c] : add;
            g[i] = add;
        }
        return;
    }
    const double inv_r2 = inv_r * inv_r;
    const double coef_qs = -2.0 * alpha * s_am1 * inv_r2;
    const double s_am2 = s_am1 * inv_s;
    const double H_ss = -alpha * (alpha - 1.0) * s_am2 * inv_r + alpha * alpha * (s_am1 * s_am1) * inv_r2 + mu * inv_s * inv_s;
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            double v = 0.0;
            const bool iq = i < nz - 1, jq = j < nz - 1;
            const bool is = i == nz - 1, js = j == nz - 1;
            if (iq && jq) v = 4.0 * z[i] * z[j] * inv_r2 + (i == j ? 2.0 * inv_r : 0.0);
            else if (iq && js) v = coef_qs * z[i];
            else if (is && jq) v = coef_qs * z[j];
            else if (is && js) v = H_ss;
            Hz[i + MI * j] = v;
        }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
        int ki = -1;
#pragma unroll
        for (int c = 0; c < MI; ++c) ki = (c < nz && P.idx[c] == i) ? c : ki;
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            int kj = -1;
#pragma unroll
            for (int c = 0; c < MI; ++c) kj = (c < nz && P.idx[c] == j) ? c : kj;
            double v = 0.0;
            if (ki >= 0 && kj >= 0) {
#pragma unroll
                for (int q = 0; q < MI * MI; ++q) v = (q == ki + MI * kj) ? Hz[q] : v;
            }
            H[i * NY + j] = v;
        }
    }
}

// Full node barrier: value F, gradient g, Hessian H for the Convex or its phase-I wrapper.
template <int NY, int ORDER>
__device__ __forceinline__ void cone_eval(const ConeDev& C, int64_t node, int64_t n, const double (&y)[NY],
                                          double& F, double (&g)[NY], double (&H)[NY * NY]) {
    if (C.npieces == 1 && !C.feasibility && C.pc[0].kind == MGBHIP_KIND_EP && C.pc[0].A == nullptr &&
        C.pc[0].b == nullptr && C.pc[0].select == nullptr) {
        ep_identity_eval<NY, ORDER>(C.pc[0], node, y, F, g, H);
        return;
    }
    F = 0.0;
    if (ORDER == 1) {
#pragma unroll
        for (int i = 0; i < NY; ++i) g[i] = 0.0;
    }
    if (ORDER == 2) {
#pragma unroll
        for (int i = 0; i < NY * NY; ++i) H[i] = 0.0;
    }
    const int slack_pos = C.feasibility ? C.NC - 1 : -1;
    for (int k = 0; k < C.npieces; ++k) {
        const PieceDev& P = C.pc[k];
        if (P.select != nullptr && P.select[node] == 0.0) continue;   // exact zero, never 0 * Inf
        piece_accumulate<NY, ORDER>(P, node, n, y, slack_pos, F, g, H);
    }
    if (C.feasibility) {
        const int NC = C.NC;
        const double bb = C.box_b, RR = C.box_R;
        const double u = pick<NY>(y, NC - 1);
        if (ORDER == 0) {
            F += -mgb_Log(bb - u) - mgb_Log(bb + u);
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < NY; ++i) acc += (i >= NC) ? (-mgb_Log(RR - y[i]) - mgb_Log(RR + y[i])) : 0.0;
            F += acc;
        } else if (ORDER == 1) {
#pragma unroll
            for (int i = 0; i < NY; ++i) {
                if (i == NC - 1) g[i] += 1.0 / (bb - u) - 1.0 / (bb + u);
                else if (i >= NC) g[i] = 1.0 / (RR - y[i]) - 1.0 / (RR + y[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NY; ++i) {
                if (i == NC - 1) H[i * NY + i] += 1.0 / ((bb - u) * (bb - u)) + 1.0 / ((bb + u) * (bb + u));
                else if (i >= NC) H[i * NY + i] = 1.0 / ((RR - y[i]) * (RR - y[i])) + 1.0 / ((RR + y[i]) * (RR + y[i]));
            }
        }
    }
}

// Slack initialiser (src/convex_euclidian_power.jl:243-253, src/convex_linear.jl:205-214,
// src/convex_piecewise.jl:62-75): max over active pieces.
template <int NY>
__device__ __forceinline__ double cone_slack(const ConeDev& C, int64_t node, int64_t n, const double (&y)[NY]) {
    constexpr int MI = MGBHIP_MAX_IDX;
    double out = -INFINITY;
    bool any = false;
    for (int k = 0; k < C.npieces; ++k) {
        const PieceDev& P = C.pc[k];
        if (P.select != nullptr && P.select[node] == 0.0) continue;
        const int ni = P.ni;
        double yk[MI];
#pragma unroll
        for (int c = 0; c < MI; ++c) yk[c] = (c < ni) ? pick<NY>(y, P.idx[c]) : 0.0;
        double val;
        const int nr = (P.kind == MGBHIP_KIND_EP) ? ni : P.nc;
        double z[MI];
#pragma unroll
        for (int r = 0; r < MI; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < MI; ++c) {
                double a = 0.0;
                if (r < nr && c < ni) a = P.A ? P.A[node + n * (int64_t)(r + nr * c)] : (r == c ? 1.0 : 0.0);
                acc += a * yk[c];
            }
            z[r] = acc + ((P.b != nullptr && r < nr) ? P.b[node + n * (int64_t)r] : 0.0);
        }
        if (P.kind == MGBHIP_KIND_EP) {
            const double p0 = P.p ? P.p[node] : P.p_const;
            double qsq = 0.0;
#pragma unroll
            for (int r = 0; r < MI; ++r) qsq += (r < ni - 1) ? z[r] * z[r] : 0.0;
            const double s = pick<MI>(z, ni - 1);
            val = -fmin(s - mgb_safe_pow(qsq, p0 / 2.0), s);
        } else {
            double mn = INFINITY;
#pragma unroll
            for (int r = 0; r < MI; ++r) mn = (r < nr) ? fmin(mn, z[r]) : mn;
            val = -mn;
        }
        out = any ? fmax(out, val) : val;
        any = true;
    }
    return out;
}

}  // namespace mgbhip
