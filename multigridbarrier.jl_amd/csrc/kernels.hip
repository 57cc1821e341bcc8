// kernels.hip -- evaluate / assemble kernels of the barrier functional on gfx950.
//
// One fused element kernel family replaces the reference's chain
//   R*s (SpMV) -> apply_D (nD block matvecs) -> map_rows_gpu(F) -> D' back-multiplies ->
//   16 fused triple products + _hess_add! temporaries
// (reference: src/convex.jl:155-202, src/BlockMatrices.jl:170-188, :604-640; CUDA twins
// ext/MultiGridBarrierCUDAExt/block_ops.jl:31-148, map_rows_gpu.jl:20-28): a group of
// G = 2^ceil(log2 p) lanes owns one element, lane r owns node r.  The element's operator
// blocks are staged through LDS with a flat coalesced copy (they are contiguous in the
// reference's p x p x N layout), Dz, the cone functor and the per-element 14x14 (nu*p)
// Hessian block never leave the CU.  HBM traffic is the compulsory one of SURVEY.md
// section 8(d): operators + z + grids in, element blocks out.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>

#include "kernels.hpp"
#include "dense.hpp"

namespace mgbhip {

namespace {

// Sum of one double per thread over a 256-thread workgroup, result in thread 0: rows of 16 lanes on the data-parallel
// path (v_mov dpp: quad_perm 0xB1 / 0x4E, row_ror 4 / 8), the four rows of a wave through the crossbar, the four waves
// through LDS -- one barrier instead of the eight of an LDS tree (the tail of every element-kernel workgroup).
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// z0 + R s at row i.  Selection levels: the same arithmetic as prolong_kernel with a unit entry (v = z0; v += 1 * s[col]).
__device__ __forceinline__ double z_at(const ElemParams& P, int64_t i) {
    double v = P.z0[i];
    if (P.zsel) {
        const int32_t c = P.zsel[i];
        if (c >= 0) v += P.zx ? __builtin_fma(-P.zalpha, P.zx[c], P.zs[c]) : P.zs[c];      // the same fma as step_kernel
        if (P.zout) P.zout[i] = v;
    }
    return v;
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_mov_f64<0xB1>(v);
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0x124>(v);
    v += dpp_mov_f64<0x128>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ double block_sum_256(double v, double* scratch4) {      // scratch4: 4 doubles of LDS, free to use
    v = wave_sum_dpp(v);
    if ((threadIdx.x & 63) == 0) scratch4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch4[0] + scratch4[1]) + (scratch4[2] + scratch4[3]);
}

__device__ __forceinline__ int tri_index(int k, int k2, int NY) {   // k <= k2
    return k * NY - (k * (k - 1)) / 2 + (k2 - k);
}

template <int NY, int MODE>
__global__ __launch_bounds__(256) void elem_kernel(const ElemParams P, const int lgG) {
    extern __shared__ double sh[];
    const int tid = threadIdx.x;
    const int G = 1 << lgG;
    const int EPB = 256 >> lgG;
    const int el = tid >> lgG;
    const int r = tid & (G - 1);
    const int p = P.p;
    const int pp = p * p;
    const int nu = P.nu;
    const int64_t e = (int64_t)blockIdx.x * EPB + el;
    const bool active = (e < P.N) && (r < p);
    const int64_t n = P.n;
    const int64_t node = e * p + r;

    double* zl = sh;                                    // [EPB][nu][G]
    double* opL = zl + 256 * nu;                        // [nstage][EPB][pp]
    double* YL = opL + (size_t)P.nstage * EPB * pp;     // MODE_F1: [EPB][NY][G]; MODE_F2: [EPB][tri][G]

    // 1. stage the operator blocks of this workgroup's elements (flat, coalesced)
    {
        const int64_t e0 = (int64_t)blockIdx.x * EPB;
        int64_t lim = (P.N - e0) * pp;
        if (lim > (int64_t)EPB * pp) lim = (int64_t)EPB * pp;
        for (int o = 0; o < P.nstage; ++o) {
            const double* src = P.stage_ptr[o] + e0 * pp;
            double* dst = opL + (size_t)o * EPB * pp;
            for (int i = tid; i < lim; i += 256) dst[i] = src[i];
        }
    }
    // 2. fine broken-basis values of this element: z0 + R*s  (src/convex.jl:156)
    if (active) {
        for (int a = 0; a < nu; ++a) zl[(el * nu + a) * G + r] = z_at(P, (int64_t)a * n + node);
    }
    __syncthreads();

    auto OP = [&](int k, int rr, int cc) -> double {    // D_k block entry (rr, cc) of this element
        const int so = P.D_stage[k];
        if (so >= 0) return opL[((size_t)so * EPB + el) * pp + cc * p + rr];
        return P.ops[P.D_op[k]][e * pp + cc * p + rr];
    };

    // 3. Dz at this node (src/convex.jl:125)
    double y[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) {
        double v = 0.0;
        if (active) {
            const int a = P.D_state[k];
            if (P.D_stage[k] == -1) {
                v = zl[(el * nu + a) * G + r];
            } else {
                for (int cc = 0; cc < p; ++cc) v += OP(k, r, cc) * zl[(el * nu + a) * G + cc];
            }
        }
        y[k] = v;
    }

    double F = 0.0;
    double g[NY];
    double H[NY * NY];
    (void)g;
    (void)H;

    if (MODE == MODE_F0 || MODE == MODE_NODE_F) {
        if (active) cone_eval<NY, 0>(P.cone, node, n, y, F, g, H);
        if (MODE == MODE_NODE_F) {
            if (active) {
                P.out_F[node] = F;
                if (P.out_Dz != nullptr) {
#pragma unroll
                    for (int k = 0; k < NY; ++k) P.out_Dz[node + n * k] = y[k];
                }
            }
            return;
        }
        double val = 0.0;
        if (active) {
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
        __syncthreads();            // zl / opL no longer needed: reuse LDS for the reduction
        const double tot = block_sum_256(val, sh);
        if (tid == 0) P.out_partial[blockIdx.x] = tot;
        return;
    }
    if (MODE == MODE_NODE_SLACK) {
        if (active) P.out_F[node] = cone_slack<NY>(P.cone, node, n, y);
        return;
    }
    if (MODE == MODE_F01) {
        // One line-search trial (src/newton.jl:35-50 evaluates F0 then F1 at the same point): the operator
        // blocks, z and c are streamed once instead of twice.
        double val = 0.0;
        if (active) {
            double F0v;
            cone_eval<NY, 0>(P.cone, node, n, y, F0v, g, H);
            cone_eval<NY, 1>(P.cone, node, n, y, F, g, H);
            const double wv = P.w[node];
            const double bwv = P.bw ? P.bw[node] : 0.0;
            const double bar = P.bw ? ((bwv == 0.0) ? 0.0 : bwv * F0v) : P.invn * F0v;
            double lin = 0.0;
#pragma unroll
            for (int k = 0; k < NY; ++k) {
                const double ck = P.c[node + n * k];
                lin += ck * y[k];
                const double sc = P.bw ? ((bwv == 0.0) ? 0.0 : bwv * g[k]) : P.invn * g[k];
                YL[(el * NY + k) * G + r] = sc + wv * ck;
            }
            val = bar + wv * lin;
        }
        __syncthreads();
        if (active) {
            const int i = r;
            for (int a = 0; a < nu; ++a) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NY; ++k) {
                    if (P.D_state[k] != a) continue;
                    const double* Yk = YL + (el * NY + k) * G;
                    if (P.D_stage[k] == -1) {
                        acc += Yk[i];
                    } else {
                        for (int rr = 0; rr < p; ++rr) acc += OP(k, rr, i) * Yk[rr];
                    }
                }
                P.out_ret[(int64_t)a * n + node] = acc;
            }
        }
        __syncthreads();            // zl / opL / YL no longer needed: reuse LDS for the reduction
        const double tot = block_sum_256(val, sh);
        if (tid == 0) P.out_partial[blockIdx.x] = tot;
        return;
    }
    if (MODE == MODE_F1) {
        // Y = scale(grad F) + w .* c   (src/convex.jl:170-173), then sum_k D_k' Y_k per element
        if (active) {
            cone_eval<NY, 1>(P.cone, node, n, y, F, g, H);
            const double wv = P.w[node];
            const double bwv = P.bw ? P.bw[node] : 0.0;
#pragma unroll
            for (int k = 0; k < NY; ++k) {
                double sc = P.bw ? ((bwv == 0.0) ? 0.0 : bwv * g[k]) : P.invn * g[k];
                YL[(el * NY + k) * G + r] = sc + wv * P.c[node + n * k];
            }
        }
        __syncthreads();
        if (active) {
            const int i = r;
            for (int a = 0; a < nu; ++a) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NY; ++k) {
                    if (P.D_state[k] != a) continue;
                    const double* Yk = YL + (el * NY + k) * G;
                    if (P.D_stage[k] == -1) {
                        acc += Yk[i];
                    } else {
                        for (int rr = 0; rr < p; ++rr) acc += OP(k, rr, i) * Yk[rr];
                    }
                }
                P.out_ret[(int64_t)a * n + node] = acc;
            }
        }
        return;
    }
    if (MODE == MODE_F2) {
        constexpr int NT = NY * (NY + 1) / 2;
        if (active) {
            cone_eval<NY, 2>(P.cone, node, n, y, F, g, H);
            const double bwv = P.bw ? P.bw[node] : 0.0;
#pragma unroll
            for (int k = 0; k < NY; ++k)
#pragma unroll
                for (int k2 = k; k2 < NY; ++k2) {
                    const double h = H[k * NY + k2];
                    const double sc = P.bw ? ((bwv == 0.0) ? 0.0 : bwv * h) : P.invn * h;
                    YL[((size_t)el * NT + tri_index(k, k2, NY)) * G + r] = sc;
                }
        }
        __syncthreads();
        if (active) {
            // lane j = r owns column j of every block (a,b), a <= b, of the element Hessian
            //   Hel_ab[i,j] = sum_rr sum_{k in K_a} sum_{k2 in K_b} D_k[rr,i] Y[rr][k,k2] D_k2[rr,j]
            // (src/convex.jl:191-200 with the 16 temporaries fused away)
            const int j = r;
            const int NB = nu * (nu + 1) / 2;
            for (int a = 0; a < nu; ++a)
                for (int b = a; b < nu; ++b) {
                    const int blk = a * nu - (a * (a - 1)) / 2 + (b - a);
                    const bool dblk = (P.diag_mask >> blk) & 1;
                    double* out = P.out_hel + P.blk_off[blk] + (dblk ? (e * p + j) - j : (e * p + j) * (int64_t)p);
                    for (int i = dblk ? j : 0; i < (dblk ? j + 1 : p); ++i) {
                        double val = 0.0;
#pragma unroll
                        for (int k = 0; k < NY; ++k) {
                            if (P.D_state[k] != a) continue;
                            const bool idk = P.D_stage[k] == -1;
#pragma unroll
                            for (int k2 = 0; k2 < NY; ++k2) {
                                if (P.D_state[k2] != b) continue;
                                const bool idk2 = P.D_stage[k2] == -1;
                                const int t = (k <= k2) ? tri_index(k, k2, NY) : tri_index(k2, k, NY);
                                const double* Yt = YL + ((size_t)el * NT + t) * G;
                                if (idk && idk2) {
                                    val += (i == j) ? Yt[i] : 0.0;
                                } else if (idk) {
                                    val += Yt[i] * OP(k2, i, j);
                                } else if (idk2) {
                                    val += OP(k, j, i) * Yt[j];
                                } else {
                                    double acc = 0.0;
                                    for (int rr = 0; rr < p; ++rr) acc += OP(k, rr, i) * Yt[rr] * OP(k2, rr, j);
                                    val += acc;
                                }
                            }
                        }
                        out[i] = val;
                    }
                }
        }
        return;
    }
}

// Specialised element Hessian kernel for compile-time (NY, P): same arithmetic as MODE_F2 of the
// generic kernel, restructured so that lane j first forms C_k[r] = sum_k' Y_r[k,k'] D_k'[r,j] in
// registers and then out[i] = sum_k sum_r D_k[r,i] C_k[r]  (|K_a| * P * (|K_b| + P) multiply-adds
// per block instead of |K_a| |K_b| P^2), all loops unrolled, operators and Y in LDS, the
// finished blocks staged through LDS and written with a flat coalesced copy (the slab is
// block-major: [block][element][P*P]).
// D-table signatures.  SigRuntime reads the (state, operator slot) rows from the kernel
// arguments; SigDefault<NY> is the reference's default_D layout (src/mgb.jl:595-607)
//   [u id; u dx; (u dy; (u dz;)) s id]  with the default cone idx = 2:dim+2,
// i.e. rows 1..NY-1 enter the barrier, row 0 (u itself) does not.  With a compile-time
// signature every set-membership test below folds away and the block products shrink to the
// structurally non-zero terms.
struct SigRuntime {
    static constexpr bool rt = true;
    static __device__ __forceinline__ constexpr int state(int) { return 0; }
    static __device__ __forceinline__ constexpr int stage(int) { return 0; }
    static __device__ __forceinline__ constexpr int mask() { return 0; }
};
template <int NY>
struct SigDefault {
    static constexpr bool rt = false;
    static __device__ __forceinline__ constexpr int state(int k) { return k == NY - 1 ? 1 : 0; }
    static __device__ __forceinline__ constexpr int stage(int k) { return (k == 0 || k == NY - 1) ? -1 : k - 1; }
    static __device__ __forceinline__ constexpr int mask() { return ((1 << NY) - 1) & ~1; }
};

template <int NY, int P, class Sig, bool CONDENSE = false>
__global__ __launch_bounds__(256) void elem_f2_fast(const ElemParams Pm) {
    constexpr int G = (P <= 2) ? 2 : (P <= 4) ? 4 : (P <= 8) ? 8 : (P <= 16) ? 16 : (P <= 32) ? 32 : 64;
    constexpr int EPB = 256 / G;
    constexpr int PP = P * P;
    constexpr int NT = NY * (NY + 1) / 2;
    extern __shared__ double sh[];
    const int tid = threadIdx.x;
    const int el = tid / G;
    const int r = tid % G;
    const int nu = Sig::rt ? Pm.nu : 2;
    auto DST = [&](int k) -> int { return Sig::rt ? Pm.D_state[k] : Sig::state(k); };
    auto DSG = [&](int k) -> int { return Sig::rt ? Pm.D_stage[k] : Sig::stage(k); };
    const int ymask = Sig::rt ? Pm.ymask : Sig::mask();
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int64_t e = e0 + el;
    const bool active = (e < Pm.N) && (r < P);
    const int64_t n = Pm.n;
    const int64_t node = e * P + r;

    double* zl = sh;                                    // [EPB][nu][G]
    double* opL = zl + 256 * nu;                        // [nstage][EPB][PP]
    double* YL = opL + (size_t)Pm.nstage * EPB * PP;    // [EPB][NT][G]

    // z0 + R s of this lane's node: requested first (selection levels chain two loads: column, then s)
    double zr[MGBHIP_MAX_NU];
#pragma unroll
    for (int a = 0; a < MGBHIP_MAX_NU; ++a) zr[a] = (active && a < nu) ? z_at(Pm, (int64_t)a * n + node) : 0.0;
    {   // operator blocks of this workgroup's elements -> LDS.  All loads of a stage are issued before the first
        // LDS store (a rolled copy loop waits one memory latency per iteration)
        int64_t lim = (Pm.N - e0) * PP;
        if (lim > (int64_t)EPB * PP) lim = (int64_t)EPB * PP;
        constexpr int NIT = (EPB * PP + 255) / 256;
        for (int o = 0; o < Pm.nstage; ++o) {
            const double* src = Pm.stage_ptr[o] + e0 * PP;
            double* dst = opL + (size_t)o * EPB * PP;
            double v[NIT];
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int i = tid + 256 * u;
                v[u] = i < lim ? src[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int i = tid + 256 * u;
                if (i < lim) dst[i] = v[u];
            }
        }
    }
    if (active) {
#pragma unroll
        for (int a = 0; a < MGBHIP_MAX_NU; ++a)
            if (a < nu) zl[(el * nu + a) * G + r] = zr[a];
    }
    __syncthreads();
    const double* opE = opL + (size_t)el * PP;           // + slot * EPB * PP
    auto OP = [&](int k, int rr, int cc) -> double { return opE[(size_t)DSG(k) * EPB * PP + cc * P + rr]; };

    double y[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) {
        double v = 0.0;
        if (active) {
            const double* za = zl + (el * nu + DST(k)) * G;
            if (DSG(k) < 0) v = za[r];
            else {
#pragma unroll
                for (int cc = 0; cc < P; ++cc) v += OP(k, r, cc) * za[cc];
            }
        }
        y[k] = v;
    }
    if (active) {
        double F, g[NY], H[NY * NY];
        cone_eval<NY, 2>(Pm.cone, node, n, y, F, g, H);
        const double bwv = Pm.bw ? Pm.bw[node] : 0.0;
#pragma unroll
        for (int k = 0; k < NY; ++k)
#pragma unroll
            for (int k2 = k; k2 < NY; ++k2) {
                const double h = H[k * NY + k2];
                YL[((size_t)el * NT + (k * NY - (k * (k - 1)) / 2 + (k2 - k))) * G + r] =
                    Pm.bw ? ((bwv == 0.0) ? 0.0 : bwv * h) : Pm.invn * h;
            }
    }
    __syncthreads();
    const int j = r;
    const double* Ye = YL + (size_t)el * NT * G;
    int blk = 0;
    double cblk[CONDENSE ? 3 : 1][P];      // CONDENSE: column j of the uu / us blocks and ss_j stay in registers
#pragma unroll
    for (int a = 0; a < nu; ++a)
#pragma unroll
        for (int b = a; b < nu; ++b, ++blk) {
            // lane j owns column j of the block: P contiguous doubles of the block-major slab; a wave
            // covers 64/G whole blocks, so every cache line is completed within the wave's stores
            const bool dblk = CONDENSE ? (blk == 2) : ((Pm.diag_mask >> blk) & 1);      // diagonal block, stored compactly
            double* dst = CONDENSE ? &cblk[blk < 3 ? blk : 0][0]
                                   : Pm.out_hel + Pm.blk_off[blk] + (dblk ? (e * P + j) : (e * P + j) * (int64_t)P);
            bool b_all_id = true;
#pragma unroll
            for (int k2 = 0; k2 < NY; ++k2)
                if (DST(k2) == b && ((ymask >> k2) & 1) && DSG(k2) >= 0) b_all_id = false;
            if (active && b_all_id) {
                // every operator of state b is the identity: C_k[r] = delta(r, j) * sum_k' Y_j[k,k']
                double Cd[NY];
#pragma unroll
                for (int k = 0; k < NY; ++k) {
                    double acc = 0.0;
                    if (DST(k) == a && ((ymask >> k) & 1)) {
#pragma unroll
                        for (int k2 = 0; k2 < NY; ++k2) {
                            if (DST(k2) != b || !((ymask >> k2) & 1)) continue;
                            const int t = (k <= k2) ? (k * NY - (k * (k - 1)) / 2 + (k2 - k))
                                                    : (k2 * NY - (k2 * (k2 - 1)) / 2 + (k - k2));
                            acc += Ye[t * G + j];
                        }
                    }
                    Cd[k] = acc;
                }
                if (dblk) {          // state a carries identity operators only as well
                    double val = 0.0;
#pragma unroll
                    for (int k = 0; k < NY; ++k)
                        if (DST(k) == a && ((ymask >> k) & 1)) val += Cd[k];
                    dst[0] = val;
                } else {
#pragma unroll
                    for (int i = 0; i < P; ++i) {
                        double val = 0.0;
#pragma unroll
                        for (int k = 0; k < NY; ++k) {
                            if (DST(k) != a || !((ymask >> k) & 1)) continue;
                            if (DSG(k) < 0) val += (i == j) ? Cd[k] : 0.0;
                            else val += OP(k, j, i) * Cd[k];
                        }
                        dst[i] = val;
                    }
                }
            } else if (active) {
                double C[NY][P];
#pragma unroll
                for (int k = 0; k < NY; ++k) {
                    if (DST(k) != a || !((ymask >> k) & 1)) continue;
#pragma unroll
                    for (int rr = 0; rr < P; ++rr) {
                        double acc = 0.0;
#pragma unroll
                        for (int k2 = 0; k2 < NY; ++k2) {
                            if (DST(k2) != b || !((ymask >> k2) & 1)) continue;
                            const int t = (k <= k2) ? (k * NY - (k * (k - 1)) / 2 + (k2 - k))
                                                    : (k2 * NY - (k2 * (k2 - 1)) / 2 + (k - k2));
                            const double yv = Ye[t * G + rr];
                            if (DSG(k2) < 0) acc += (rr == j) ? yv : 0.0;
                            else acc += yv * OP(k2, rr, j);
                        }
                        C[k][rr] = acc;
                    }
                }
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    double val = 0.0;
#pragma unroll
                    for (int k = 0; k < NY; ++k) {
                        if (DST(k) != a || !((ymask >> k) & 1)) continue;
                        if (DSG(k) < 0) val += C[k][i];
                        else {
#pragma unroll
                            for (int rr = 0; rr < P; ++rr) val += OP(k, rr, i) * C[k][rr];
                        }
                    }
                    dst[i] = val;
                }
            }
        }
    if constexpr (CONDENSE) {
        // ---- partial factorization of the element's leaf front (kernels.hpp: launch_elem_f2_condense) --------------
        // Leaf index list: [slack of node 0..P-1 | interior u node (element node P-1) | the other element nodes that
        // are unknowns, in the front's order | border].  Lane j holds uu(:, j), us(:, j) (u_i against slack j), ss_j.
        static_assert(!Sig::rt && P <= 8, "condensation: default two-state signature only");
        constexpr int PB = P - 1;                     // the interior node
        __syncthreads();                              // every lane is done with the staged operators: reuse their LDS
        double* X = opL + (size_t)el * (2 * PP);      // per-element scratch (nstage == 2: 2 * PP doubles per element)
        double* Xus = X;                              // [P][P]: Xus[q * P + i] = us(i, q)
        double* Xinv = X + PP;                        // [P] 1 / ss_q
        double* Xbeta = Xinv + P;                     // [P] border entries -g of the slacks
        double* Xc = Xbeta + P;                       // [P] uu'(:, PB) after the slack elimination
        double* Xs = Xc + P;                          // [0] border entry of the interior node after the slacks, [1] its pivot
        const double* uuc = cblk[0];
        const double* usc = cblk[1];
        const double ssj = cblk[2][0];
        LeafDesc ld{0, 0, 0};
        double inv = 0.0, beta = 0.0;
        bool bad = false;
        if (active) {
            ld = Pm.leaf_desc[e];
            beta = -Pm.leaf_g[Pm.leaf_slack0 + node];
            bad = (ssj == 0.0) || !isfinite(ssj);
            inv = 1.0 / ssj;
#pragma unroll
            for (int i = 0; i < P; ++i) Xus[j * P + i] = usc[i];
            Xinv[j] = inv;
            Xbeta[j] = beta;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double up[P];                                 // uu'(i, j) = uu(i, j) - sum_q us(i, q) us(j, q) / ss_q
        double bj = 0.0, corner = 0.0;
        if (active) {
            double t[P];
#pragma unroll
            for (int q = 0; q < P; ++q) t[q] = Xus[q * P + j] * Xinv[q];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                double acc = uuc[i];
#pragma unroll
                for (int q = 0; q < P; ++q) acc -= Xus[q * P + i] * t[q];
                up[i] = acc;
            }
#pragma unroll
            for (int q = 0; q < P; ++q) bj -= Xbeta[q] * t[q];                 // border row entry of u_j after the slacks
            if (j == PB) {
                bj -= Pm.leaf_g[ld.interior];                                   // the interior node is a pivot of this leaf
#pragma unroll
                for (int q = 0; q < P; ++q) corner -= Xbeta[q] * Xbeta[q] * Xinv[q];
#pragma unroll
                for (int i = 0; i < P; ++i) Xc[i] = up[i];
                Xs[0] = bj;
                Xs[1] = up[PB];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            // The finished leaf front: square column-major (entry (r, c) at r + c*m) straight to the arena, or -- packed
            // leaves -- the lower triangle (column c at c*m - c(c-1)/2) staged in LDS and copied out in one coalesced
            // run of m(m+1)/2 doubles per element.
            const int m = (int)(ld.packed & 15u);
            const bool pk = Pm.leaf_packed != 0;
            int pos[P];
#pragma unroll
            for (int i = 0; i < P; ++i) pos[i] = (int)((ld.packed >> (4 + 4 * i)) & 15u);
            double d7 = 1.0, bb = 0.0, inv7 = 1.0, f = 0.0, xc[P];
#pragma unroll
            for (int i = 0; i < P; ++i) xc[i] = 0.0;
            if (active) {
                d7 = Xs[1];
                bb = Xs[0];
                inv7 = 1.0 / d7;
#pragma unroll
                for (int i = 0; i < P; ++i) xc[i] = Xc[i];
                f = (j < PB) ? xc[j] * inv7 : 0.0;
                bad = bad || (j == PB && ((d7 == 0.0) || !isfinite(d7)));
            }
            double* Fg = Pm.leaf_arena + ld.F_off;
            double* S = Fg;                               // destination of the scattered writes
            if (pk) {
                __syncthreads();                          // every lane has read its scratch: the staging area may overlap it
                S = sh + (size_t)el * 120;
            }
            auto at = [&](int rr, int cc) -> int { return pk ? cc * m - (cc * (cc - 1)) / 2 + (rr - cc) : rr + cc * m; };
            if (active) {
                // slack column j: pivot, zeros against the later slacks, the u rows, the border row
                S[at(j, j)] = ssj;
#pragma unroll
                for (int rr = 0; rr < P; ++rr)
                    if (rr > j) S[at(rr, j)] = 0.0;
                S[at(P, j)] = usc[PB] * inv;
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    if (pos[i] != 15) S[at(pos[i], j)] = usc[i] * inv;
                S[at(m - 1, j)] = beta * inv;
                if (j == PB) {                             // column of the interior node
                    S[at(P, P)] = d7;
#pragma unroll
                    for (int i = 0; i < PB; ++i)
                        if (pos[i] != 15) S[at(pos[i], P)] = xc[i] * inv7;
                    S[at(m - 1, P)] = bb * inv7;
                    S[at(m - 1, m - 1)] = corner - bb * bb * inv7;
                } else if (pos[j] != 15) {                 // update column of element node j
#pragma unroll
                    for (int i = 0; i < PB; ++i)
                        if (pos[i] != 15 && pos[i] >= pos[j]) S[at(pos[i], pos[j])] = up[i] - xc[i] * f;
                    S[at(m - 1, pos[j])] = bj - bb * f;
                }
            }
            if (pk) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (e < Pm.N) {
                    const LeafDesc l2 = Pm.leaf_desc[e];       // lane G-1 of the group takes part in the copy
                    const int mm = (int)(l2.packed & 15u);
                    double* dstF = Pm.leaf_arena + l2.F_off;
                    for (int t = r; t < mm * (mm + 1) / 2; t += G) dstF[t] = S[t];
                }
            }
        }
        if (bad) atomicOr(Pm.leaf_status, 1);
    }
}

// Specialised line-search trial (MODE_F01 of the generic kernel: value and gradient at one point from one
// pass over the operator blocks) for compile-time (NY, P) and D-table signature: every loop unrolled, the
// set-membership tests of the D table folded away.  Same arithmetic, same summation order.
template <int NY, int P, class Sig>
__global__ __launch_bounds__(256) void elem_f01_fast(const ElemParams Pm) {
    constexpr int G = (P <= 2) ? 2 : (P <= 4) ? 4 : (P <= 8) ? 8 : (P <= 16) ? 16 : (P <= 32) ? 32 : 64;
    constexpr int EPB = 256 / G;
    constexpr int PP = P * P;
    extern __shared__ double sh[];
    const int tid = threadIdx.x;
    const int el = tid / G;
    const int r = tid % G;
    const int nu = Sig::rt ? Pm.nu : 2;
    auto DST = [&](int k) -> int { return Sig::rt ? Pm.D_state[k] : Sig::state(k); };
    auto DSG = [&](int k) -> int { return Sig::rt ? Pm.D_stage[k] : Sig::stage(k); };
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int64_t e = e0 + el;
    const bool active = (e < Pm.N) && (r < P);
    const int64_t n = Pm.n;
    const int64_t node = e * P + r;

    double* zl = sh;                                    // [EPB][nu][G]
    double* opL = zl + 256 * nu;                        // [nstage][EPB][PP]
    double* YL = opL + (size_t)Pm.nstage * EPB * PP;    // [EPB][NY][G]
    {   // operator blocks of this workgroup's elements -> LDS.  All loads of a stage are issued before the first
        // LDS store (a rolled copy loop waits one memory latency per iteration)
        int64_t lim = (Pm.N - e0) * PP;
        if (lim > (int64_t)EPB * PP) lim = (int64_t)EPB * PP;
        constexpr int NIT = (EPB * PP + 255) / 256;
        for (int o = 0; o < Pm.nstage; ++o) {
            const double* src = Pm.stage_ptr[o] + e0 * PP;
            double* dst = opL + (size_t)o * EPB * PP;
            double v[NIT];
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int i = tid + 256 * u;
                v[u] = i < lim ? src[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int i = tid + 256 * u;
                if (i < lim) dst[i] = v[u];
            }
        }
    }
    // the node's cost row and weights do not depend on anything staged: request them with the operators
    double ck[NY];
    double wv = 0.0, bwv = 0.0;
    if (active) {
        for (int a = 0; a < nu; ++a) zl[(el * nu + a) * G + r] = z_at(Pm, (int64_t)a * n + node);
#pragma unroll
        for (int k = 0; k < NY; ++k) ck[k] = Pm.c[node + n * k];
        wv = Pm.w[node];
        bwv = Pm.bw ? Pm.bw[node] : 0.0;
    }
    __syncthreads();
    const double* opE = opL + (size_t)el * PP;
    auto OP = [&](int k, int rr, int cc) -> double { return opE[(size_t)DSG(k) * EPB * PP + cc * P + rr]; };
    double y[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) {
        double v = 0.0;
        if (active) {
            const double* za = zl + (el * nu + DST(k)) * G;
            if (DSG(k) < 0) v = za[r];
            else {
#pragma unroll
                for (int cc = 0; cc < P; ++cc) v += OP(k, r, cc) * za[cc];
            }
        }
        y[k] = v;
    }
    double val = 0.0;
    if (active) {
        double F0v, F, g[NY], H[NY * NY];
        cone_eval<NY, 0>(Pm.cone, node, n, y, F0v, g, H);
        cone_eval<NY, 1>(Pm.cone, node, n, y, F, g, H);
        const double bar = Pm.bw ? ((bwv == 0.0) ? 0.0 : bwv * F0v) : Pm.invn * F0v;
        double lin = 0.0;
#pragma unroll
        for (int k = 0; k < NY; ++k) {
            lin += ck[k] * y[k];
            const double sc = Pm.bw ? ((bwv == 0.0) ? 0.0 : bwv * g[k]) : Pm.invn * g[k];
            YL[(el * NY + k) * G + r] = sc + wv * ck[k];
        }
        val = bar + wv * lin;
    }
    __syncthreads();
    if (active) {
        const int i = r;
        for (int a = 0; a < nu; ++a) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < NY; ++k) {
                if (DST(k) != a) continue;
                const double* Yk = YL + (el * NY + k) * G;
                if (DSG(k) < 0) {
                    acc += Yk[i];
                } else {
#pragma unroll
                    for (int rr = 0; rr < P; ++rr) acc += OP(k, rr, i) * Yk[rr];
                }
            }
            Pm.out_ret[(int64_t)a * n + node] = acc;
        }
    }
    __syncthreads();            // operators / Y no longer needed: reuse LDS for the reduction
    const double tot = block_sum_256(val, sh);
    if (tid == 0) Pm.out_partial[blockIdx.x] = tot;
}

// ---- compensated sums --------------------------------------------------------------------------
// The coarse-level sums run over every element of the mesh, and near the end of a barrier solve one node's term can
// exceed the rest by sixteen orders of magnitude (an iterate 1e-13 from the cone's wall: Hessian entries ~ 1e16, soft
// eigenvalues ~ 1e1).  A plain running sum then loses the other 10^5 terms below the ulp of the large one -- an
// absolute error of hundreds in a matrix whose smallest eigenvalue is 34: H comes out indefinite by summation
// noise alone (tests/dev/logs/gpu_coarse_noise_probe_L8_p1.5.txt).  TwoSum accumulation (Knuth) carries the rounding
// error of every addition in a second word: the sum is the correctly rounded one to a few ulps, whatever the order.
// Used by the gather_assemble_* kernels and the long-row restrictions.  (At cond(H) ~ 1e15 the sign of lambda^2 also
// depends on the rounding of the per-node terms themselves, which no summation scheme removes: DESIGN.md section 5.)
struct DSum {
    double s = 0.0, c = 0.0;
    __device__ __forceinline__ void add(double x) {
#ifdef MGB_PLAIN_SUMS
        s += x;
        return;
#endif
        const double t = s + x;
        const double bp = t - s;
        c += (s - (t - bp)) + (x - bp);
        s = t;
    }
    __device__ __forceinline__ void merge(double s2, double c2) { add(s2); c += c2; }
    __device__ __forceinline__ double value() const { return s + c; }
};
__device__ __forceinline__ void dsum_wave_reduce(DSum& a) {      // fixed shuffle tree over the 64 lanes; result in lane 0
    for (int off = 32; off > 0; off >>= 1) {
        const double s2 = __shfl_down(a.s, off, 64), c2 = __shfl_down(a.c, off, 64);
        a.merge(s2, c2);
    }
}

// ---- reductions ------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ partials, int64_t count,
                                                              double* __restrict__ out) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int64_t i = tid; i < count; i += 256) s += partials[i];
    red[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) out[0] = red[0];
}

// MODE 0: sum a*b ; MODE 1: sum a*a and count of non-finite a
template <int MODE>
__global__ __launch_bounds__(256) void block_reduce_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                           int64_t n, double* __restrict__ partials,
                                                           const double* __restrict__ mask) {
    __shared__ double red[256];
    __shared__ double red2[256];
    const int tid = threadIdx.x;
    double s = 0.0, bad = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = a[i];
        const double w = mask ? mask[i] : 1.0;       // domain decomposition: 1 on the entries this rank owns, else 0
        if (MODE == 0) s += mask ? w * (v * b[i]) : v * b[i];
        else {
            s += mask ? w * (v * v) : v * v;
            bad += isfinite(v) ? 0.0 : 1.0;
        }
    }
    red[tid] = s;
    red2[tid] = bad;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { red[tid] += red[tid + off]; red2[tid] += red2[tid + off]; }
        __syncthreads();
    }
    if (tid == 0) {
        partials[blockIdx.x] = red[0];
        if (MODE == 1) partials[gridDim.x + blockIdx.x] = red2[0];
    }
}

// Line-search trial, everything behind the element kernel in ONE launch: the restriction g = R' ret (row gather, as
// csr_matvec_row_kernel), the partial sums of |g|^2 and its non-finite count (the same grid-stride order and LDS tree as
// block_reduce_kernel<1>: identical partials), and the step kernel's work -- xn = x - s n with its own fused multiply-add and
// the "moved" stamp.  Three launches fewer per trial than restrict + block_reduce + step.
__global__ __launch_bounds__(256) void restrict_trial_kernel(int64_t rows, const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                             const double* __restrict__ val, const double* __restrict__ ret,
                                                             double* __restrict__ g, double* __restrict__ partials,
                                                             const double* __restrict__ x, const double* __restrict__ nn, double s,
                                                             double* __restrict__ xn, int32_t* __restrict__ moved, int32_t stamp) {
    __shared__ double red[256];
    __shared__ double red2[256];
    __shared__ int any_moved;
    const int tid = threadIdx.x;
    if (tid == 0) any_moved = 0;
    __syncthreads();
    double ss = 0.0, bad = 0.0;
    bool m = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < rows; i += (int64_t)gridDim.x * 256) {
        double acc = 0.0;
        for (int32_t q = ptr[i]; q < ptr[i + 1]; ++q) acc += val[q] * ret[col[q]];
        g[i] = acc;
        ss += acc * acc;
        bad += isfinite(acc) ? 0.0 : 1.0;
        if (xn) {
            const double xi = x[i];
            const double v = __builtin_fma(-s, nn[i], xi);
            xn[i] = v;
            m = m || (v != xi);
        }
    }
    red[tid] = ss;
    red2[tid] = bad;
    if (m) any_moved = 1;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { red[tid] += red[tid + off]; red2[tid] += red2[tid + off]; }
        __syncthreads();
    }
    if (tid == 0) {
        partials[blockIdx.x] = red[0];
        partials[gridDim.x + blockIdx.x] = red2[0];
        if (any_moved) *moved = stamp;
    }
}

// Newton direction statistics in one pass: sum v*v, count of non-finite v, and g.v (the same per-block partial sums
// and trees as block_reduce_kernel<1> and <0>: identical values, two launches fewer per Newton iteration)
__global__ __launch_bounds__(256) void dir_stats_kernel(const double* __restrict__ v, const double* __restrict__ g, int64_t n,
                                                        double* __restrict__ partials, const double* __restrict__ mask) {
    __shared__ double red[3][256];
    const int tid = threadIdx.x;
    double s = 0.0, bad = 0.0, d = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = v[i];
        const double w = mask ? mask[i] : 1.0;
        s += mask ? w * (x * x) : x * x;
        bad += isfinite(x) ? 0.0 : 1.0;
        d += mask ? w * (g[i] * x) : g[i] * x;
    }
    red[0][tid] = s; red[1][tid] = bad; red[2][tid] = d;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            red[0][tid] += red[0][tid + off];
            red[1][tid] += red[1][tid + off];
            red[2][tid] += red[2][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        partials[blockIdx.x] = red[0][0];
        partials[gridDim.x + blockIdx.x] = red[1][0];
        partials[2 * gridDim.x + blockIdx.x] = red[2][0];
    }
}

// Second stage of every two-stage reduction, ONE launch for everything the host wants from one synchronisation:
// up to four strided sums (each: 256 threads stride over the partials + the same LDS tree as before, so the values
// are bit for bit those of the separate reduce_partials / reduce2 launches this replaces), device flags that ride
// behind the sums as doubles (a pivot status, the step kernel's "moved" stamp; `reset` clears them for their next
// producer: no hipMemsetAsync per Newton iteration), and a copy of out[host_lo .. host_lo + host_n) straight into
// the pinned host block (host-coherent memory: visible after the stream synchronises; no copy launch).
struct FinishJob { const double* src; int64_t count; int32_t out; };
struct FinishParams {
    FinishJob job[4];
    int32_t njobs;
    double* out;
    int32_t* ints;
    int32_t nints, ints_out, reset;
    double* host;
    int32_t host_lo, host_n;
    double seq;                  // != 0: written to host[15] after the results (system-scope fence in between): the host polls it
};
__global__ __launch_bounds__(256) void finish_kernel(const FinishParams P) {
    __shared__ double red[256];
    __shared__ double res[16];
    const int tid = threadIdx.x;
    if (tid < P.nints) {
        res[P.ints_out + tid] = (double)P.ints[tid];
        if (P.reset) P.ints[tid] = 0;
    }
    for (int o = 0; o < P.njobs; ++o) {
        const double* __restrict__ src = P.job[o].src;
        const int64_t cnt = P.job[o].count;
        double s = 0.0;
        for (int64_t i = tid; i < cnt; i += 256) s += src[i];
        red[tid] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) red[tid] += red[tid + off];
            __syncthreads();
        }
        if (tid == 0) res[P.job[o].out] = red[0];
        __syncthreads();
    }
    __syncthreads();
    // results -> device scalar block (every slot this launch produced) and -> host
    if (tid < 16) {
        bool mine = false;
        for (int o = 0; o < P.njobs; ++o) mine = mine || (P.job[o].out == tid);
        if (tid >= P.ints_out && tid < P.ints_out + P.nints) mine = true;
        if (mine) P.out[tid] = res[tid];
        if (P.host && tid >= P.host_lo && tid < P.host_lo + P.host_n) P.host[tid] = mine ? res[tid] : P.out[tid];
    }
    if (P.host && P.seq != 0.0) {
        __threadfence_system();                      // this thread's result stores are visible to the host ...
        __syncthreads();                             // ... for every writing thread, before the stamp
        if (tid == 0) {
            __hip_atomic_store(P.host + 15, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- sparse matvecs ------------------------------------------------------------------------------

template <bool ADD>
__global__ __launch_bounds__(256) void csr_matvec_row_kernel(int64_t rows, const int32_t* __restrict__ ptr,
                                                             const int32_t* __restrict__ col,
                                                             const double* __restrict__ val,
                                                             const double* __restrict__ x, double* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    double s = 0.0;
    for (int32_t q = ptr[i]; q < ptr[i + 1]; ++q) s += val[q] * x[col[q]];
    y[i] = ADD ? y[i] + s : s;
}

template <bool ADD>
__global__ __launch_bounds__(256) void csr_matvec_wave_kernel(int64_t rows, const int32_t* __restrict__ ptr,
                                                              const int32_t* __restrict__ col,
                                                              const double* __restrict__ val,
                                                              const double* __restrict__ x, double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    DSum a;
    for (int32_t q = ptr[i] + lane; q < ptr[i + 1]; q += 64) a.add(val[q] * x[col[q]]);
    dsum_wave_reduce(a);
    if (lane == 0) y[i] = ADD ? y[i] + a.value() : a.value();
}

// Very long rows (restriction onto a handful of coarse unknowns: every row of R' spans a large
// part of the mesh): a workgroup per (row, 4096-entry chunk), partial sums to scratch, then a
// fixed-order sum over the chunks -- deterministic, and the whole GPU works on a 3-row matvec.
constexpr int CHUNK = 4096;
__global__ __launch_bounds__(256) void csr_matvec_chunk_kernel(const int32_t* __restrict__ ptr,
                                                               const int32_t* __restrict__ col,
                                                               const double* __restrict__ val,
                                                               const double* __restrict__ x,
                                                               double* __restrict__ partial, int nchunk) {
    __shared__ double red[256];
    const int row = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int32_t q0 = ptr[row] + ch * CHUNK;
    const int32_t q1 = min(ptr[row + 1], q0 + CHUNK);
    __shared__ double redc[256];
    DSum a;
    for (int32_t q = q0 + tid; q < q1; q += 256) a.add(val[q] * x[col[q]]);
    red[tid] = a.s;
    redc[tid] = a.c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            DSum b;
            b.s = red[tid]; b.c = redc[tid];
            b.merge(red[tid + off], redc[tid + off]);
            red[tid] = b.s; redc[tid] = b.c;
        }
        __syncthreads();
    }
    if (tid == 0) {            // (sum, carried error) of the chunk
        partial[2 * ((int64_t)row * nchunk + ch)] = red[0];
        partial[2 * ((int64_t)row * nchunk + ch) + 1] = redc[0];
    }
}

// one wave per row: the lanes stride over the chunk sums (a row of the coarsest level has 224 of them at L = 9; one thread
// walking them serially took 19 us for a 2-row matvec), fixed shuffle tree
__global__ __launch_bounds__(256) void csr_matvec_chunk_sum_kernel(int64_t rows, const double* __restrict__ partial,
                                                                   int nchunk, double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    DSum a;
    for (int c = lane; c < nchunk; c += 64) a.merge(partial[2 * (i * nchunk + c)], partial[2 * (i * nchunk + c) + 1]);
    dsum_wave_reduce(a);
    if (lane == 0) y[i] = a.value();
}

// zfull = z0 + R*s  (src/convex.jl:156): one thread per broken row, so the element kernels
// read their local values with independent coalesced loads instead of a dependent
// rowptr -> col/val -> s chain per tile.
__global__ __launch_bounds__(256) void prolong_kernel(int64_t rows, const int32_t* __restrict__ ptr,
                                                      const int32_t* __restrict__ col,
                                                      const double* __restrict__ val,
                                                      const double* __restrict__ s, const double* __restrict__ z0,
                                                      double* __restrict__ zfull) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    double v = z0[i];
    for (int32_t q = ptr[i]; q < ptr[i + 1]; ++q) v += val[q] * s[col[q]];
    zfull[i] = v;
}

__global__ __launch_bounds__(256) void step_kernel(const double* __restrict__ x, const double* __restrict__ nn,
                                                   double s, double* __restrict__ xn, int64_t len,
                                                   int32_t* __restrict__ moved, int32_t stamp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool m = false;
    if (i < len) {
        const double xi = x[i];
        const double v = __builtin_fma(-s, nn[i], xi);       // one rounding (what the contraction of xi - s * nn[i] gave): z_at forms the same value on the fly
        xn[i] = v;
        m = (v != xi);
    }
    // one store per workgroup at most, and only when something moved (atomics on one word
    // from every wave serialise at the memory side)
    __shared__ int any_moved;
    if (threadIdx.x == 0) any_moved = 0;
    __syncthreads();
    if (m) any_moved = 1;
    __syncthreads();
    // the flag carries the caller's stamp of THIS step (a fresh value per launch): nobody has to clear it beforehand
    if (threadIdx.x == 0 && any_moved) *moved = stamp;
}

__global__ __launch_bounds__(256) void scale_copy_kernel(const double* __restrict__ src, double alpha,
                                                         double* __restrict__ dst, int64_t len) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < len) dst[i] = alpha * src[i];
}

// border column of the bordered Newton system: tail[0 .. m) = -g, tail[m] = -1 (one launch)
__global__ __launch_bounds__(256) void border_tail_kernel(const double* __restrict__ g, double* __restrict__ tail, int64_t m) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < m) tail[i] = -1.0 * g[i];
    else if (i == m) tail[i] = -1.0;
}

__global__ __launch_bounds__(256) void axpy_kernel(double alpha, const double* __restrict__ x,
                                                   double* __restrict__ y, int64_t len) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < len) y[i] += alpha * x[i];
}

__global__ __launch_bounds__(256) void fill_kernel(double value, double* __restrict__ y, int64_t len) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < len) y[i] = value;
}

// ---- assembly ------------------------------------------------------------------------------------

// Row-owner gather: every structural nonzero of R'HR sums its contributions from the
// element-block slab in a fixed order -- no atomics (the reference's CUDA path uses fp64
// atomics, ext/MultiGridBarrierCUDAExt/block_ops.jl:229-249).
// cidx == nullptr: the slab is already in list order (projected levels: the projection kernels scatter through
// PanelParams::spos, so a list is a contiguous run and the gather streams it).
// qmap (optional): the positions to assemble -- the Newton loop forms the UPPER triangle only (`symmetric(H)` and the
// factorization read nothing else, mf_analysis.cpp), nq of the nnz structural nonzeros.
__global__ __launch_bounds__(256) void gather_assemble_kernel(int64_t nq, const int32_t* __restrict__ qmap,
                                                              const int32_t* __restrict__ cptr,
                                                              const int32_t* __restrict__ cidx,
                                                              const double* __restrict__ slab,
                                                              double* __restrict__ Hval) {
    const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    const int64_t q = qmap ? qmap[qi] : qi;
    const int32_t beg = cptr[q], end = cptr[q + 1];
    if (end - beg <= 4) {              // a handful of element contributions: nothing to compensate
        double s = 0.0;
        for (int32_t t = beg; t < end; ++t) s += slab[cidx ? cidx[t] : t];
        Hval[q] = s;
        return;
    }
    DSum a;
    for (int32_t t = beg; t < end; ++t) a.add(slab[cidx ? cidx[t] : t]);
    Hval[q] = a.value();
}

// Direct-value levels: only the structural nonzeros shared between elements are summed (into a compact array
// behind the slab); the single-contribution ones are read from the slab by the factorization itself.
__global__ __launch_bounds__(256) void gather_shared_kernel(int64_t nshared, const int32_t* __restrict__ sh_q,
                                                            const int32_t* __restrict__ cptr, const int32_t* __restrict__ cidx,
                                                            const double* __restrict__ slab, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nshared) return;
    const int32_t q = sh_q[i];
    const int32_t beg = cptr[q], end = cptr[q + 1];
    if (end - beg <= 4) {                                    // same rule and order as gather_assemble_kernel: bitwise the same sums
        double a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = (beg + u < end) ? slab[cidx[beg + u]] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (beg + u < end) s += a[u];
        out[i] = s;
        return;
    }
    DSum acc;
    for (int32_t t = beg; t < end; t += 4) {                 // four contributions in flight, added in list order
        double a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = (t + u < end) ? slab[cidx[t + u]] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t + u < end) acc.add(a[u]);
    }
    out[i] = acc.value();
}

// Long contribution lists (coarse levels: few unknowns, every element contributes): one wave per
// structural nonzero, lanes stride over the list, fixed-order shuffle reduction.
__global__ __launch_bounds__(256) void gather_assemble_wave_kernel(int64_t nq, const int32_t* __restrict__ qmap,
                                                                   const int32_t* __restrict__ cptr,
                                                                   const int32_t* __restrict__ cidx,
                                                                   const double* __restrict__ slab,
                                                                   double* __restrict__ Hval) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const int64_t q = qmap ? qmap[qi] : qi;
    DSum a;
    for (int32_t t = cptr[q] + lane; t < cptr[q + 1]; t += 64) a.add(slab[cidx ? cidx[t] : t]);
    dsum_wave_reduce(a);
    if (lane == 0) Hval[q] = a.value();
}

// Very long lists (the coarsest levels: a handful of nonzeros, each summing every element): one wave per
// (nonzero, chunk of the list), then one wave per nonzero over the chunk sums.  Fixed chunking and fixed
// shuffle trees: the result does not depend on scheduling.
__global__ __launch_bounds__(256) void gather_assemble_chunk_kernel(int64_t nq, const int32_t* __restrict__ qmap, int32_t ch,
                                                                    int32_t nchunk,
                                                                    const int32_t* __restrict__ cptr,
                                                                    const int32_t* __restrict__ cidx,
                                                                    const double* __restrict__ slab,
                                                                    double* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nq * nchunk) return;
    const int64_t qi = w / nchunk;
    const int32_t c = (int32_t)(w - qi * nchunk);
    const int64_t q = qmap ? qmap[qi] : qi;
    const int32_t beg = cptr[q] + c * ch, end = min(cptr[q + 1], beg + ch);
    DSum a;
    for (int32_t t = beg + lane; t < end; t += 256) {      // four loads in flight per lane
        const int32_t t1 = t + 64, t2 = t + 128, t3 = t + 192;
        const double a0 = slab[cidx ? cidx[t] : t];
        const double a1 = t1 < end ? slab[cidx ? cidx[t1] : t1] : 0.0;
        const double a2 = t2 < end ? slab[cidx ? cidx[t2] : t2] : 0.0;
        const double a3 = t3 < end ? slab[cidx ? cidx[t3] : t3] : 0.0;
        a.add(a0); a.add(a1); a.add(a2); a.add(a3);
    }
    dsum_wave_reduce(a);
    if (lane == 0) { part[2 * w] = a.s; part[2 * w + 1] = a.c; }
}

__global__ __launch_bounds__(256) void gather_assemble_chunk_reduce(int64_t nq, const int32_t* __restrict__ qmap, int32_t nchunk,
                                                                    const double* __restrict__ part, double* __restrict__ Hval) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    DSum a;
    for (int32_t c = lane; c < nchunk; c += 64) a.merge(part[2 * (qi * nchunk + c)], part[2 * (qi * nchunk + c) + 1]);
    dsum_wave_reduce(a);
    if (lane == 0) Hval[qmap ? qmap[qi] : qi] = a.value();
}

// General (coarse) levels: one wave per element computes the projected block
// [panel_0 .. panel_{nu-1}]' * Hel_e * [panel_0 .. panel_{nu-1}] in two steps per block pair
// (tmp = Hel_ab * panel_b in LDS, then panel_a' * tmp) into the element's slab; the structural
// nonzeros of H gather from the slab afterwards (deterministic, no atomics).
__global__ __launch_bounds__(256) void panel_project_kernel(const PanelParams P) {
    extern __shared__ double sh[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 4 + wave;
    if (e >= P.N) return;
    const int p = P.p, nu = P.nu;
    const int NB2 = nu * (nu + 1) / 2;
    double* tmp = sh + (size_t)wave * p * P.cmax;
    const int32_t base = P.ecol_ptr[e * nu];
    const int32_t ct = P.ecol_ptr[(e + 1) * nu] - base;
    for (int a = 0; a < nu; ++a) {
        const int32_t oa = P.ecol_ptr[e * nu + a], ca = P.ecol_ptr[e * nu + a + 1] - oa;
        const double* pa = P.panels + (int64_t)p * oa;
        for (int b = P.upper_only ? a : 0; b < nu; ++b) {        // upper_only: block pairs below the diagonal are never read
            const int32_t ob = P.ecol_ptr[e * nu + b], cb = P.ecol_ptr[e * nu + b + 1] - ob;
            const double* pb = P.panels + (int64_t)p * ob;
            const bool tr = a > b;
            const int blk = tr ? (b * nu - (b * (b - 1)) / 2 + (a - b)) : (a * nu - (a * (a - 1)) / 2 + (b - a));
            const double* Hb = P.hel + ((int64_t)blk * P.N + e) * (int64_t)p * p;
            for (int t = lane; t < p * cb; t += 64) {
                const int rr = t % p, ib = t / p;
                double acc = 0.0;
                for (int ss = 0; ss < p; ++ss) acc += (tr ? Hb[ss + p * rr] : Hb[rr + p * ss]) * pb[ss + p * ib];
                tmp[t] = acc;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int t = lane; t < ca * cb; t += 64) {
                const int ia = t % ca, ib = t / ca;
                if (P.upper_only && a == b && ia > ib) continue;
                double acc = 0.0;
                for (int rr = 0; rr < p; ++rr) acc += pa[rr + p * ia] * tmp[rr + p * ib];
                const int64_t o = P.eoff[e] + (oa - base + ia) + (int64_t)ct * (ob - base + ib);
                P.slab[P.spos ? P.spos[o] : o] = acc;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// Small coarse levels whose basis functions overlap almost everywhere (3-D hierarchies): a per-element
// slab would hold sum_e ct_e^2 doubles for an m x m system of a few hundred unknowns, and the gather
// behind it reads them back at random.  Instead a workgroup owns a stream of elements and a chunk of the
// packed upper triangle of H as an LDS accumulator: per element the panels, blocks and T = Hel * P are
// staged in LDS, the projected entries that fall into the chunk are added in place (no global
// read-modify-write, no atomics), and at the end the chunk goes to the stream's partial result.  A
// second kernel sums the streams in order.  Deterministic; traffic = inputs + nstream * m^2 / 2 doubles.
__global__ __launch_bounds__(256) void panel_accumulate_kernel(const PanelParams P, const int32_t* __restrict__ ecols,
                                                               int32_t m, int32_t ctmax, int32_t chunk,
                                                               double* __restrict__ partial) {
    extern __shared__ double sh[];
    const int tid = threadIdx.x;
    const int nstream = gridDim.x, stream = blockIdx.x;
    const int p = P.p, nu = P.nu;
    const int nblk = nu * (nu + 1) / 2;
    const int64_t mt = (int64_t)m * (m + 1) / 2;       // packed upper triangle: (gi <= gj) at gi + gj (gj + 1) / 2
    const int64_t lo = (int64_t)blockIdx.y * chunk;
    const int64_t hi = lo + chunk < mt ? lo + chunk : mt;
    double* acc = sh;                                  // [chunk]
    double* Pl = acc + chunk;                          // Pl[rr + p*j]
    double* Hl = Pl + (size_t)p * ctmax;               // Hl[blk*p*p + rr + p*ss]
    double* Tl = Hl + (size_t)nblk * p * p;            // Tl[(a*p + rr) + nu*p*j]
    int32_t* cl = reinterpret_cast<int32_t*>(Tl + (size_t)nu * p * ctmax);   // cl[j] column, cl[ctmax + j] state
    for (int t = tid; t < chunk; t += 256) acc[t] = 0.0;
    const int nrow = nu * p;
    // The staging data of the NEXT element travel in registers while the current one is processed
    // (launch_panel_accumulate guarantees p*ctmax <= 4*256, nblk*p*p <= 3*256, ctmax <= 256).
    double rP[4], rH[3];
    int32_t rC = 0, rS = 0, nbase = 0, nct = 0;
    auto prefetch = [&](int64_t e) {
        nbase = P.ecol_ptr[e * nu];
        nct = P.ecol_ptr[(e + 1) * nu] - nbase;
        if (tid < nct) {
            rC = ecols[nbase + tid];
            int st = 0;
            for (int a = 1; a < nu; ++a)
                if (nbase + tid >= P.ecol_ptr[e * nu + a]) st = a;
            rS = st;
        }
        const double* pan = P.panels + (int64_t)p * nbase;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = tid + 256 * k;
            rP[k] = t < p * nct ? pan[t] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int t = tid + 256 * k;
            if (t < nblk * p * p) {
                const int blk = t / (p * p), q = t - blk * (p * p);
                rH[k] = P.hel[((int64_t)blk * P.N + e) * (int64_t)(p * p) + q];
            }
        }
    };
    if (stream < P.N) prefetch(stream);
    for (int64_t e = stream; e < P.N; e += nstream) {
        const int32_t ct = nct;
        __syncthreads();                               // previous element's LDS operands are no longer read
        if (tid < ct) { cl[tid] = rC; cl[ctmax + tid] = rS; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = tid + 256 * k;
            if (t < p * ct) Pl[t] = rP[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int t = tid + 256 * k;
            if (t < nblk * p * p) Hl[t] = rH[k];
        }
        __syncthreads();
        if (e + nstream < P.N) prefetch(e + nstream);  // in flight during the two product phases below
        // columns whose packed positions can fall into this workgroup's chunk [lo, hi)
        int jlo, jhi;
        {
            int a0 = 0, a1 = ct;                       // first column whose largest position reaches lo
            while (a0 < a1) {
                const int mid = (a0 + a1) >> 1;
                const int64_t g = cl[mid];
                if (g + g * (g + 1) / 2 < lo) a0 = mid + 1; else a1 = mid;
            }
            jlo = a0;
            a0 = jlo; a1 = ct;                         // first column whose smallest position is >= hi
            const int64_t g0 = cl[0];
            while (a0 < a1) {
                const int mid = (a0 + a1) >> 1;
                const int64_t g = cl[mid];
                if (g0 + g * (g + 1) / 2 < hi) a0 = mid + 1; else a1 = mid;
            }
            jhi = a0;
        }
        const float inv_nrow = 1.0f / (float)nrow;
        for (int t = tid + nrow * jlo; t < nrow * jhi; t += 256) {   // T[a][rr][j] = sum_ss Hel_{a, b(j)}[rr][ss] * P[ss][j], j in [jlo, jhi)
            int j = (int)(((float)t + 0.5f) * inv_nrow);
            if (j * nrow > t) --j;
            if ((j + 1) * nrow <= t) ++j;
            const int row = t - j * nrow;
            const int a = row / p, rr = row - a * p;
            const int b = cl[ctmax + j];
            const bool tr = a > b;
            const int blk = tr ? (b * nu - (b * (b - 1)) / 2 + (a - b)) : (a * nu - (a * (a - 1)) / 2 + (b - a));
            const double* Hb = Hl + (size_t)blk * p * p;
            double v = 0.0;
            for (int ss = 0; ss < p; ++ss) v += (tr ? Hb[ss + p * rr] : Hb[rr + p * ss]) * Pl[ss + p * j];
            Tl[row + nrow * j] = v;
        }
        __syncthreads();
        // B[i][j] for the pairs i <= j (the element's columns are sorted, so gi <= gj) whose packed
        // position falls into this workgroup's chunk.  Positions grow with t = i + j (j + 1) / 2, so
        // the chunk is a contiguous t range: bracket it by columns, then enumerate only that range.
        const int tbeg = jlo * (jlo + 1) / 2, tend = jhi * (jhi + 1) / 2;
        for (int t = tbeg + tid; t < tend; t += 256) {
            int j = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while ((j + 1) * (j + 2) / 2 <= t) ++j;
            while (j * (j + 1) / 2 > t) --j;
            const int i = t - j * (j + 1) / 2;
            const int64_t gi = cl[i], gj = cl[j];
            const int64_t pos = gi + gj * (gj + 1) / 2;
            if (pos < lo || pos >= hi) continue;
            const int a = cl[ctmax + i];
            double v = 0.0;
            for (int rr = 0; rr < p; ++rr) v += Pl[rr + p * i] * Tl[(a * p + rr) + nrow * j];
            acc[pos - lo] += v;                        // distinct (gi, gj) per thread within an element
        }
    }
    __syncthreads();
    double* out = partial + (int64_t)stream * mt + lo;
    for (int t = tid; t < (int)(hi - lo); t += 256) out[t] = acc[t];
}

// H[i, j] = H[j, i] = sum over the streams' partial results, in stream order (i <= j)
__global__ __launch_bounds__(256) void accumulate_reduce_kernel(int32_t m, int32_t nstream, const double* __restrict__ partial,
                                                                double* __restrict__ H) {
    const int64_t mt = (int64_t)m * (m + 1) / 2;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= (int64_t)m * m) return;
    const int i = (int)(q % m), j = (int)(q / m);
    if (i > j) return;
    const double* src = partial + (int64_t)i + ((int64_t)j * (j + 1)) / 2;
    double s = 0.0;
    int w = 0;
    for (; w + 8 <= nstream; w += 8) {          // eight independent loads in flight, fixed summation order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(w + u) * mt];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; w < nstream; ++w) s += src[(int64_t)w * mt];
    H[(int64_t)i * m + j] = s;
    H[(int64_t)j * m + i] = s;
}

// Slab variant of the same staging (levels that keep the slab + gather path): the element's
// panels, blocks and T = Hel * P live in LDS, the ct x ct projected block is written with flat
// coalesced stores.  Replaces the per-block-pair loops of panel_project_kernel, which re-read the
// panels from L2 for every output entry.
__global__ __launch_bounds__(256) void panel_project_staged_kernel(const PanelParams P, int32_t ctmax) {
    extern __shared__ double sh[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 4 + wave;
    if (e >= P.N) return;
    const int p = P.p, nu = P.nu;
    const int nblk = nu * (nu + 1) / 2;
    const size_t per_wave = (size_t)p * ctmax + (size_t)nblk * p * p + (size_t)nu * p * ctmax + ctmax;
    double* Pl = sh + (size_t)wave * per_wave;
    double* Hl = Pl + (size_t)p * ctmax;
    double* Tl = Hl + (size_t)nblk * p * p;
    int32_t* sl = reinterpret_cast<int32_t*>(Tl + (size_t)nu * p * ctmax);   // sl[j]: state of column j
    const int32_t base = P.ecol_ptr[e * nu];
    const int32_t ct = P.ecol_ptr[(e + 1) * nu] - base;
    for (int j = lane; j < ct; j += 64) {
        int st = 0;
        for (int a = 1; a < nu; ++a)
            if (base + j >= P.ecol_ptr[e * nu + a]) st = a;
        sl[j] = st;
    }
    const double* pan = P.panels + (int64_t)p * base;
    for (int t = lane; t < p * ct; t += 64) Pl[t] = pan[t];
    for (int blk = 0; blk < nblk; ++blk) {
        const double* hb = P.hel + ((int64_t)blk * P.N + e) * (int64_t)(p * p);
        for (int q = lane; q < p * p; q += 64) Hl[blk * p * p + q] = hb[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nrow = nu * p;
    const float inv_nrow = 1.0f / (float)nrow, inv_ct = 1.0f / (float)ct;
    for (int t = lane; t < nrow * ct; t += 64) {
        int j = (int)(((float)t + 0.5f) * inv_nrow);
        if (j * nrow > t) --j;
        if ((j + 1) * nrow <= t) ++j;
        const int row = t - j * nrow;
        const int a = row / p, rr = row - a * p;
        const int b = sl[j];
        const bool tr = a > b;
        const int blk = tr ? (b * nu - (b * (b - 1)) / 2 + (a - b)) : (a * nu - (a * (a - 1)) / 2 + (b - a));
        const double* Hb = Hl + (size_t)blk * p * p;
        double acc = 0.0;
        for (int ss = 0; ss < p; ++ss) acc += (tr ? Hb[ss + p * rr] : Hb[rr + p * ss]) * Pl[ss + p * j];
        Tl[row + nrow * j] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t eo = P.eoff[e];
    if (P.upper_only) {                     // the Newton loop reads entries i <= j only (the element's columns are sorted)
        for (int t = lane; t < ct * (ct + 1) / 2; t += 64) {
            int j = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while ((j + 1) * (j + 2) / 2 <= t) ++j;
            while (j * (j + 1) / 2 > t) --j;
            const int i = t - j * (j + 1) / 2;
            const int a = sl[i];
            double acc = 0.0;
            for (int rr = 0; rr < p; ++rr) acc += Pl[rr + p * i] * Tl[(a * p + rr) + nrow * j];
            const int64_t o = eo + i + ct * j;
            P.slab[P.spos ? P.spos[o] : o] = acc;
        }
        return;
    }
    for (int t = lane; t < ct * ct; t += 64) {
        int j = (int)(((float)t + 0.5f) * inv_ct);
        if (j * ct > t) --j;
        if ((j + 1) * ct <= t) ++j;
        const int i = t - j * ct;
        const int a = sl[i];
        double acc = 0.0;
        for (int rr = 0; rr < p; ++rr) acc += Pl[rr + p * i] * Tl[(a * p + rr) + nrow * j];
        P.slab[P.spos ? P.spos[eo + t] : eo + t] = acc;                       // entry i + ct * j of the element's block
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Projection on the matrix cores (round 4).  The loop kernels above spend their time decoding flat indices and on two LDS
// reads per multiply-add; the arithmetic itself is two small dense products per element and state pair (a, b),
//     U = Hel_ab * P_b   (p x c_b)      and      B_ab = P_a' U   (c_a x c_b, K = p),
// i.e. 16 x 16 tiles of v_mfma_f64_16x16x4 with K = p padded to a multiple of 4.  The element's columns are laid out with
// every state's range padded to a multiple of 16, so a tile belongs to one state; `cmap` takes a padded column back to the
// element's compact index, -1 for padding.  One workgroup per element, its four waves share the tile pairs I <= J; U never
// touches LDS: register r of the first product's result, D[i = fk + 4 r][j = fr], IS the second product's B operand of
// k-step r (B[k = 4 r + fk][j = fr]).  LDS holds P and the element block only (13 KB at 96 padded columns: eight
// workgroups per compute unit; a first version that staged T = Hel P kept three, and the launch is latency-bound:
// processing several elements per workgroup in sequence was slower still).
// Operand convention of the instruction as used throughout this library (mf_numeric.hip): lane (fr = lane & 15,
// fk = lane >> 4) supplies A[i = fr][k = fk] and B[k = fk][j = fr]; afterwards register r holds D[i = fk + 4 r][j = fr].
typedef double pp_double4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int pp_round_up(int v, int q) { return (v + q - 1) / q * q; }

struct PanelMfmaLayout {          // LDS layout in doubles (host and device agree through this one function)
    int ppad, KP, NR, HC, oP, oH, oC, total;
};
__host__ __device__ inline PanelMfmaLayout panel_mfma_layout(int p, int nu, int ctpad) {
    PanelMfmaLayout L;
    L.ppad = pp_round_up(p, 4);
    L.KP = L.ppad + 1;                               // leading dimension of P (k fastest)
    L.NR = nu * p + 1;                               // leading dimension of the symmetric element block (row fastest)
    L.HC = nu * L.ppad;                              // its columns: state b, k padded (zeros)
    L.oP = 0;
    L.oH = L.oP + L.KP * ctpad;
    L.oC = L.oH + L.NR * L.HC;
    L.total = L.oC + (ctpad + 1) / 2 + 8;            // cmap: ctpad int32
    return L;
}

__global__ __launch_bounds__(256) void panel_project_mfma_kernel(const PanelParams P, int32_t ctpad_max) {
    extern __shared__ double sh[];
    static_assert(MGBHIP_MAX_NU == 4, "the state offsets below are written out for four states");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t e = blockIdx.x;
    const int p = P.p, nu = P.nu, pp = p * p;
    const int nblk = nu * (nu + 1) / 2;
    const PanelMfmaLayout Y = panel_mfma_layout(p, nu, ctpad_max);
    double* Pl = sh + Y.oP;
    double* Hf = sh + Y.oH;
    int32_t* cmap = reinterpret_cast<int32_t*>(sh + Y.oC);
    // compact (c) and padded (q) start of every state's column range, as scalars (an indexed array would live in scratch)
    const int32_t* ec = P.ecol_ptr + e * nu;
    const int32_t cbase = ec[0];
    const int w0 = ec[1] - cbase, w1 = nu > 1 ? ec[2] - ec[1] : 0, w2 = nu > 2 ? ec[3] - ec[2] : 0, w3 = nu > 3 ? ec[4] - ec[3] : 0;
    const int c1 = w0, c2 = c1 + w1, c3 = c2 + w2, ct = c3 + w3;
    const int q1 = pp_round_up(w0, 16), q2 = q1 + pp_round_up(w1, 16), q3 = q2 + pp_round_up(w2, 16), ctpad = q3 + pp_round_up(w3, 16);
    const int nJt = ctpad / 16;                      // ctpad <= ctpad_max by construction of the launch
    auto state_of_padded = [&](int jp) { return (jp >= q1 && nu > 1) + (jp >= q2 && nu > 2) + (jp >= q3 && nu > 3); };
    auto qoff = [&](int a) { return a == 0 ? 0 : a == 1 ? q1 : a == 2 ? q2 : q3; };
    auto coff = [&](int a) { return a == 0 ? 0 : a == 1 ? c1 : a == 2 ? c2 : c3; };
    auto wid = [&](int a) { return a == 0 ? w0 : a == 1 ? w1 : a == 2 ? w2 : w3; };
    // ---- stage: P in gather form (zeros in the padding), the symmetric element block (zero padding columns), cmap -----
    const double* pan = P.panels + (int64_t)p * cbase;
    for (int t = tid; t < Y.KP * ctpad; t += 256) {
        const int jp = t / Y.KP, k = t - jp * Y.KP;
        const int a = state_of_padded(jp);
        const int ia = jp - qoff(a);
        const bool real = k < p && ia < wid(a);
        Pl[t] = real ? pan[k + p * (coff(a) + ia)] : 0.0;
        if (k == 0) cmap[jp] = ia < wid(a) ? coff(a) + ia : -1;
    }
    for (int t = tid; t < Y.NR * Y.HC; t += 256) Hf[t] = 0.0;
    __syncthreads();
    for (int t = tid; t < nblk * pp; t += 256) {
        const int blk = t / pp, q = t - blk * pp;
        const int ss = q / p, rr = q - ss * p;       // Hel_ab[rr + p ss], a <= b
        int a = 0, rem = blk;
        while (rem >= nu - a) { rem -= nu - a; ++a; }
        const int b = a + rem;
        const double v = P.hel[((int64_t)blk * P.N + e) * (int64_t)pp + q];
        Hf[(a * p + rr) + Y.NR * (b * Y.ppad + ss)] = v;
        if (a != b) Hf[(b * p + ss) + Y.NR * (a * Y.ppad + rr)] = v;
    }
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;
    const int ksteps = Y.ppad / 4;                   // <= 16 (p <= 64)
    const int64_t eo = P.eoff[e];
    // ---- tile pairs I <= J: U = Hel_ab P_J (p x 16, in the accumulators), B = P_I' U --------------------------------
    const int npair = nJt * (nJt + 1) / 2;
    for (int tile = wave; tile < npair; tile += 4) {
        int J = 0, rem = tile;
        while (rem > J) { rem -= J + 1; ++J; }
        const int I = rem;                           // I <= J
        const int a = state_of_padded(16 * I), b = state_of_padded(16 * J);
        pp_double4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int rt = 0; rt < ksteps; rt += 4) {     // 16 rows of U at a time: rows 4 rt .. 4 rt + 15 of the p (padded) rows
            pp_double4 u = {0.0, 0.0, 0.0, 0.0};
            const int urow = 4 * rt + fr;            // A operand row of U's tile
            for (int kk = 0; kk < ksteps; ++kk)
                u = __builtin_amdgcn_mfma_f64_16x16x4f64(urow < p ? Hf[(a * p + urow) + Y.NR * (b * Y.ppad + 4 * kk + fk)] : 0.0,
                                                        Pl[(4 * kk + fk) + Y.KP * (16 * J + fr)], u, 0, 0, 0);
            // u[r] = U[4 rt + fk + 4 r][j = fr]: the B operand of k-step rt + r
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rt + r < ksteps)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pl[(4 * (rt + r) + fk) + Y.KP * (16 * I + fr)], u[r], acc, 0, 0, 0);
        }
        const int cj = cmap[16 * J + fr];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = cmap[16 * I + fk + 4 * r];
            if (ci < 0 || cj < 0) continue;
            if (ci <= cj) {
                const int64_t o = eo + ci + (int64_t)ct * cj;
                P.slab[P.spos ? P.spos[o] : o] = acc[r];
            }
            if (!P.upper_only && (I != J ? true : ci > cj)) {      // the other triangle: mirror of an off-diagonal tile, or
                const int64_t o = I != J ? eo + cj + (int64_t)ct * ci : eo + ci + (int64_t)ct * cj;   // the lower half of a diagonal one
                P.slab[P.spos ? P.spos[o] : o] = acc[r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void invert_lists_kernel(const int32_t* __restrict__ cidx, int64_t total, int32_t* __restrict__ spos) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < total) spos[cidx[t]] = (int32_t)t;
}

template <int NY>
void launch_elem_ny(const ElemParams& P, int mode, int lgG, dim3 grid, size_t lds, hipStream_t st) {
    switch (mode) {
        case MODE_F0: hipLaunchKernelGGL((elem_kernel<NY, MODE_F0>), grid, dim3(256), lds, st, P, lgG); break;
        case MODE_F1: hipLaunchKernelGGL((elem_kernel<NY, MODE_F1>), grid, dim3(256), lds, st, P, lgG); break;
        case MODE_F2: hipLaunchKernelGGL((elem_kernel<NY, MODE_F2>), grid, dim3(256), lds, st, P, lgG); break;
        case MODE_NODE_F: hipLaunchKernelGGL((elem_kernel<NY, MODE_NODE_F>), grid, dim3(256), lds, st, P, lgG); break;
        case MODE_NODE_SLACK: hipLaunchKernelGGL((elem_kernel<NY, MODE_NODE_SLACK>), grid, dim3(256), lds, st, P, lgG); break;
        case MODE_F01: hipLaunchKernelGGL((elem_kernel<NY, MODE_F01>), grid, dim3(256), lds, st, P, lgG); break;
        default: throw InvalidArgument("launch_elem: bad mode");
    }
}

template <int NY>
void set_lds_attr() {
    const int big = 160 * 1024;
    (void)hipFuncSetAttribute((const void*)elem_kernel<NY, MODE_F0>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)elem_kernel<NY, MODE_F1>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)elem_kernel<NY, MODE_F2>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)elem_kernel<NY, MODE_NODE_F>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)elem_kernel<NY, MODE_NODE_SLACK>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)elem_kernel<NY, MODE_F01>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipGetLastError();
}

}  // namespace

int elem_group(int p) {
    int g = 1;
    while (g < p) g <<= 1;
    return g < 2 ? 2 : g;
}

int64_t elem_grid(int p, int64_t N) {
    if (p > 64) return dense_grid((int64_t)p * N);
    const int epb = 256 / elem_group(p);
    return (N + epb - 1) / epb;
}

size_t elem_lds_bytes(const ElemParams& P, int mode) {
    const int G = elem_group(P.p);
    const int EPB = 256 / G;
    size_t d = 256 * (size_t)P.nu + (size_t)P.nstage * EPB * P.p * P.p;
    if (mode == MODE_F1 || mode == MODE_F01) d += 256 * (size_t)P.nD;
    if (mode == MODE_F2) d += 256 * (size_t)(P.nD * (P.nD + 1) / 2);
    if (d < 256) d = 256;
    return d * sizeof(double);
}

template <int NY>
static bool is_default_signature(const ElemParams& P) {
    if (P.nu != 2 || P.nD != NY || P.ymask != (((1 << NY) - 1) & ~1)) return false;
    for (int k = 0; k < NY; ++k) {
        if (P.D_state[k] != (k == NY - 1 ? 1 : 0)) return false;
        if (P.D_stage[k] != ((k == 0 || k == NY - 1) ? -1 : k - 1)) return false;
    }
    return true;
}

template <int NY, int PN>
static bool try_f2_fast(const ElemParams& P, hipStream_t st) {
    if (P.nD != NY || P.p != PN) return false;
    for (int k = 0; k < NY; ++k)
        if (P.D_stage[k] == -2) return false;          // operators not staged: generic path
    const int G = elem_group(PN);
    const int EPB = 256 / G;
    const size_t lds = (256 * (size_t)P.nu + (size_t)P.nstage * EPB * PN * PN + (size_t)EPB * (NY * (NY + 1) / 2) * G) *
                       sizeof(double);
    if (lds > 160 * 1024) return false;
    static bool attr = [] {
        (void)hipFuncSetAttribute((const void*)elem_f2_fast<NY, PN, SigRuntime>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)elem_f2_fast<NY, PN, SigDefault<NY>>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        return true;
    }();
    (void)attr;
    if (is_default_signature<NY>(P))
        hipLaunchKernelGGL((elem_f2_fast<NY, PN, SigDefault<NY>>), dim3((unsigned)elem_grid(PN, P.N)), dim3(256), lds, st, P);
    else
        hipLaunchKernelGGL((elem_f2_fast<NY, PN, SigRuntime>), dim3((unsigned)elem_grid(PN, P.N)), dim3(256), lds, st, P);
    return true;
}

bool launch_elem_f2_condense(const ElemParams& P, hipStream_t st) {
    // fem2d_P2 with bubble, default D table: 7 nodes per element, node 6 interior (the only family specialised so far)
    constexpr int NY = 4, PN = 7;
    if (P.nD != NY || P.p != PN || P.nu != 2 || P.nstage != 2 || !is_default_signature<NY>(P)) return false;
    for (int k = 0; k < NY; ++k)
        if (P.D_stage[k] == -2) return false;
    MGB_REQUIRE(P.leaf_desc && P.leaf_arena && P.leaf_g && P.leaf_status, "condensing f2: leaf arguments missing");
    const int G = elem_group(PN);
    const int EPB = 256 / G;
    const size_t lds = (256 * (size_t)P.nu + (size_t)P.nstage * EPB * PN * PN + (size_t)EPB * (NY * (NY + 1) / 2) * G) * sizeof(double);
    static bool attr = [] {
        (void)hipFuncSetAttribute((const void*)elem_f2_fast<NY, PN, SigDefault<NY>, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        return true;
    }();
    (void)attr;
    hipLaunchKernelGGL((elem_f2_fast<NY, PN, SigDefault<NY>, true>), dim3((unsigned)elem_grid(PN, P.N)), dim3(256), lds, st, P);
    MGB_HIP_CHECK(hipGetLastError());
    return true;
}

template <int NY, int PN>
static bool try_f01_fast(const ElemParams& P, hipStream_t st) {
    if (P.nD != NY || P.p != PN) return false;
    for (int k = 0; k < NY; ++k)
        if (P.D_stage[k] == -2) return false;          // operators not staged: generic path
    const int G = elem_group(PN);
    const int EPB = 256 / G;
    size_t lds = (256 * (size_t)P.nu + (size_t)P.nstage * EPB * PN * PN + (size_t)EPB * NY * G) * sizeof(double);
    if (lds < 256 * sizeof(double)) lds = 256 * sizeof(double);
    if (lds > 160 * 1024) return false;
    static bool attr = [] {
        (void)hipFuncSetAttribute((const void*)elem_f01_fast<NY, PN, SigRuntime>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)elem_f01_fast<NY, PN, SigDefault<NY>>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        return true;
    }();
    (void)attr;
    if (is_default_signature<NY>(P))
        hipLaunchKernelGGL((elem_f01_fast<NY, PN, SigDefault<NY>>), dim3((unsigned)elem_grid(PN, P.N)), dim3(256), lds, st, P);
    else
        hipLaunchKernelGGL((elem_f01_fast<NY, PN, SigRuntime>), dim3((unsigned)elem_grid(PN, P.N)), dim3(256), lds, st, P);
    return true;
}

void launch_elem(const ElemParams& P, int mode, hipStream_t st) {
    if (P.p > 64) {      // one dense spectral element: GEMV + node kernel path (dense.hip)
        launch_dense_eval(P, mode, st);
        return;
    }
    MGB_REQUIRE(P.p >= 1 && P.p <= 64, "element kernels support 1 <= p <= 64 nodes per element");
    MGB_REQUIRE(P.nD >= 1 && P.nD <= MGBHIP_MAX_ND, "nD out of range");
    if (mode == MODE_F2) {
        // compile-time specialisations for the discretisations of the BASELINE configs
        // (fem1d, fem2d_P2 with/without bubble, fem3d Q1) and their phase-I images
        if (try_f2_fast<4, 7>(P, st) || try_f2_fast<3, 2>(P, st) || try_f2_fast<5, 8>(P, st) ||
            try_f2_fast<4, 6>(P, st) || try_f2_fast<7, 7>(P, st) || try_f2_fast<6, 2>(P, st) ||
            try_f2_fast<8, 8>(P, st) || try_f2_fast<7, 6>(P, st)) {
            MGB_HIP_CHECK(hipGetLastError());
            return;
        }
    }
    if (mode == MODE_F01) {
        if (try_f01_fast<4, 7>(P, st) || try_f01_fast<3, 2>(P, st) || try_f01_fast<5, 8>(P, st) ||
            try_f01_fast<4, 6>(P, st) || try_f01_fast<7, 7>(P, st) || try_f01_fast<6, 2>(P, st) ||
            try_f01_fast<8, 8>(P, st) || try_f01_fast<7, 6>(P, st)) {
            MGB_HIP_CHECK(hipGetLastError());
            return;
        }
    }
    static std::once_flag once;
    std::call_once(once, [] {
        set_lds_attr<1>(); set_lds_attr<2>(); set_lds_attr<3>(); set_lds_attr<4>();
        set_lds_attr<5>(); set_lds_attr<6>(); set_lds_attr<7>(); set_lds_attr<8>();
        set_lds_attr<9>(); set_lds_attr<10>();
    });
    const int G = elem_group(P.p);
    int lgG = 0;
    while ((1 << lgG) < G) ++lgG;
    const dim3 grid((unsigned)elem_grid(P.p, P.N));
    const size_t lds = elem_lds_bytes(P, mode);
    MGB_REQUIRE(lds <= 160 * 1024, "element kernel LDS budget exceeded");
    switch (P.nD) {
        case 1: launch_elem_ny<1>(P, mode, lgG, grid, lds, st); break;
        case 2: launch_elem_ny<2>(P, mode, lgG, grid, lds, st); break;
        case 3: launch_elem_ny<3>(P, mode, lgG, grid, lds, st); break;
        case 4: launch_elem_ny<4>(P, mode, lgG, grid, lds, st); break;
        case 5: launch_elem_ny<5>(P, mode, lgG, grid, lds, st); break;
        case 6: launch_elem_ny<6>(P, mode, lgG, grid, lds, st); break;
        case 7: launch_elem_ny<7>(P, mode, lgG, grid, lds, st); break;
        case 8: launch_elem_ny<8>(P, mode, lgG, grid, lds, st); break;
        case 9: launch_elem_ny<9>(P, mode, lgG, grid, lds, st); break;
        case 10: launch_elem_ny<10>(P, mode, lgG, grid, lds, st); break;
    }
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_reduce_partials(const double* partials, int64_t count, double* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, partials, count, out);
    MGB_HIP_CHECK(hipGetLastError());
}


static int reduce_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

int64_t reduce_scratch_doubles(int64_t n) { return 3 * (int64_t)reduce_blocks(n); }

__global__ __launch_bounds__(256) void index_gather_kernel(const double* __restrict__ v, const int32_t* __restrict__ idx,
                                                           int64_t cnt, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < cnt) out[i] = v[idx[i]];
}
__global__ __launch_bounds__(256) void index_scatter_kernel(const double* __restrict__ in, const int32_t* __restrict__ idx,
                                                            int64_t cnt, double* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < cnt) v[idx[i]] = in[i];
}
void launch_index_gather(const double* v, const int32_t* idx, int64_t cnt, double* out, hipStream_t st) {
    if (cnt == 0) return;
    hipLaunchKernelGGL(index_gather_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, v, idx, cnt, out);
    MGB_HIP_CHECK(hipGetLastError());
}
void launch_index_scatter(const double* in, const int32_t* idx, int64_t cnt, double* v, hipStream_t st) {
    if (cnt == 0) return;
    hipLaunchKernelGGL(index_scatter_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, in, idx, cnt, v);
    MGB_HIP_CHECK(hipGetLastError());
}

static void launch_finish(const FinishParams& F, hipStream_t st) {
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, st, F);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_vec_stats(const double* v, int64_t n, double* scratch, double* stats, hipStream_t st, const double* mask,
                      const int32_t* ints, int nints) {
    const int nb = reduce_blocks(n);
    hipLaunchKernelGGL(block_reduce_kernel<1>, dim3(nb), dim3(256), 0, st, v, (const double*)nullptr, n, scratch, mask);
    FinishParams F{};
    F.job[0] = FinishJob{scratch, nb, 0};
    F.job[1] = FinishJob{scratch + nb, nb, 1};
    F.njobs = 2;
    F.out = stats;
    F.ints = const_cast<int32_t*>(ints); F.nints = nints; F.ints_out = 2; F.reset = 0;
    launch_finish(F, st);
}

void launch_dir_stats(const double* v, const double* g, int64_t n, double* scratch, double* stats3, hipStream_t st,
                      const double* mask, const int32_t* ints, int nints) {
    const int nb = reduce_blocks(n);
    hipLaunchKernelGGL(dir_stats_kernel, dim3(nb), dim3(256), 0, st, v, g, n, scratch, mask);
    FinishParams F{};
    F.job[0] = FinishJob{scratch, nb, 0};
    F.job[1] = FinishJob{scratch + nb, nb, 1};
    F.job[2] = FinishJob{scratch + 2 * nb, nb, 2};
    F.njobs = 3;
    F.out = stats3;
    F.ints = const_cast<int32_t*>(ints); F.nints = nints; F.ints_out = 3; F.reset = 0;
    launch_finish(F, st);
}

// The Newton direction's read-back in one finishing launch: scal[2] = sum v^2, scal[3] = non-finite count, scal[4] = g.v,
// scal[5], scal[6] = the solver's status flags (read AND cleared: the next factorization / condensing f2 finds them zero
// without a memset launch); scal[2..7) also lands in the pinned host block `host` (same indices).
// n <= 16 scalars of the device block to the pinned host block (through its device pointer) with the sequence stamp behind
// them: what a hipMemcpyAsync + hipStreamSynchronize pair did with a runtime copy kernel (11 us) and the runtime's wait.
__global__ void publish_kernel(const double* __restrict__ src, int n, double* __restrict__ host, double* __restrict__ stamp, double seq) {
    const int tid = threadIdx.x;
    if (tid < n) host[tid] = src[tid];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(stamp, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void launch_publish(const double* src, int n, double* host_block, int host_lo, double seq, hipStream_t st) {
    // host_block: device pointer of the 16-double pinned block; the stamp always goes to its slot 15
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, st, src, n, host_block + host_lo, host_block + 15, seq);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_dir_finish(const double* v, const double* g, int64_t n, double* scratch, double* scal, int32_t* status2, double* host,
                       hipStream_t st, const double* mask, double seq) {
    const int nb = reduce_blocks(n);
    hipLaunchKernelGGL(dir_stats_kernel, dim3(nb), dim3(256), 0, st, v, g, n, scratch, mask);
    FinishParams F{};
    F.job[0] = FinishJob{scratch, nb, 2};
    F.job[1] = FinishJob{scratch + nb, nb, 3};
    F.job[2] = FinishJob{scratch + 2 * nb, nb, 4};
    F.njobs = 3;
    F.out = scal;
    F.ints = status2; F.nints = 2; F.ints_out = 5; F.reset = 1;
    F.host = host; F.host_lo = 2; F.host_n = 5;
    F.seq = seq;
    launch_finish(F, st);
}

// One line-search trial's read-back: scal[0] = f0 (sum of the element kernel's workgroup partials), scal[2] = |g|^2,
// scal[3] = non-finite count of g, scal[4] = the step kernel's "moved" stamp; scal[0..5) -> host.
void launch_trial_finish(const double* g, int64_t n, double* scratch, const double* f0_partials, int64_t f0_count, double* scal,
                         int32_t* moved, double* host, hipStream_t st, const double* mask, double seq, bool partials_ready) {
    const int nb = reduce_blocks(n);
    if (!partials_ready) hipLaunchKernelGGL(block_reduce_kernel<1>, dim3(nb), dim3(256), 0, st, g, (const double*)nullptr, n, scratch, mask);
    FinishParams F{};
    int nj = 0;
    if (f0_partials) F.job[nj++] = FinishJob{f0_partials, f0_count, 0};
    F.job[nj++] = FinishJob{scratch, nb, 2};
    F.job[nj++] = FinishJob{scratch + nb, nb, 3};
    F.njobs = nj;
    F.out = scal;
    F.ints = moved; F.nints = 1; F.ints_out = 4; F.reset = 0;
    F.host = host; F.host_lo = 0; F.host_n = 5;
    F.seq = seq;
    launch_finish(F, st);
}

void launch_dot(const double* a, const double* b, int64_t n, double* scratch, double* out, hipStream_t st, const double* mask) {
    const int nb = reduce_blocks(n);
    hipLaunchKernelGGL(block_reduce_kernel<0>, dim3(nb), dim3(256), 0, st, a, b, n, scratch, mask);
    FinishParams F{};
    F.job[0] = FinishJob{scratch, nb, 0};
    F.njobs = 1;
    F.out = out;
    launch_finish(F, st);
}

void launch_csr_matvec(int64_t rows, const int32_t* ptr, const int32_t* col, const double* val,
                       const double* x, double* y, bool add, bool long_rows, hipStream_t st) {
    if (rows == 0) return;
    if (long_rows) {
        const dim3 grid((unsigned)((rows + 3) / 4));
        if (add) hipLaunchKernelGGL(csr_matvec_wave_kernel<true>, grid, dim3(256), 0, st, rows, ptr, col, val, x, y);
        else hipLaunchKernelGGL(csr_matvec_wave_kernel<false>, grid, dim3(256), 0, st, rows, ptr, col, val, x, y);
    } else {
        const dim3 grid((unsigned)((rows + 255) / 256));
        if (add) hipLaunchKernelGGL(csr_matvec_row_kernel<true>, grid, dim3(256), 0, st, rows, ptr, col, val, x, y);
        else hipLaunchKernelGGL(csr_matvec_row_kernel<false>, grid, dim3(256), 0, st, rows, ptr, col, val, x, y);
    }
    MGB_HIP_CHECK(hipGetLastError());
}

int csr_chunks(int64_t max_row_len) { return (int)((max_row_len + CHUNK - 1) / CHUNK); }

void launch_restrict_trial(int64_t rows, const int32_t* ptr, const int32_t* col, const double* val, const double* ret, double* g,
                            double* scratch, const double* x, const double* nn, double s, double* xn, int32_t* moved, int32_t stamp,
                            hipStream_t st) {
    if (rows == 0) return;
    const int nb = reduce_blocks(rows);                      // the partial-sum layout launch_trial_finish reads
    hipLaunchKernelGGL(restrict_trial_kernel, dim3(nb), dim3(256), 0, st, rows, ptr, col, val, ret, g, scratch, x, nn, s, xn, moved, stamp);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_csr_matvec_chunked(int64_t rows, const int32_t* ptr, const int32_t* col, const double* val,
                               const double* x, double* y, double* scratch, int nchunk, hipStream_t st) {
    if (rows == 0) return;
    hipLaunchKernelGGL(csr_matvec_chunk_kernel, dim3((unsigned)nchunk, (unsigned)rows), dim3(256), 0, st, ptr, col, val, x,
                       scratch, nchunk);
    hipLaunchKernelGGL(csr_matvec_chunk_sum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, rows, scratch,
                       nchunk, y);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_prolong(int64_t rows, const int32_t* ptr, const int32_t* col, const double* val, const double* s,
                    const double* z0, double* zfull, hipStream_t st) {
    if (rows == 0) return;
    hipLaunchKernelGGL(prolong_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, rows, ptr, col, val, s,
                       z0, zfull);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_step(const double* x, const double* n, double s, double* xn, int64_t len, int32_t* moved, int32_t stamp,
                 hipStream_t st) {
    if (len == 0) return;
    hipLaunchKernelGGL(step_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, x, n, s, xn, len, moved, stamp);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_scale_copy(const double* src, double alpha, double* dst, int64_t len, hipStream_t st) {
    if (len == 0) return;
    hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, src, alpha, dst, len);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_border_tail(const double* g, double* tail, int64_t m, hipStream_t st) {
    hipLaunchKernelGGL(border_tail_kernel, dim3((unsigned)((m + 1 + 255) / 256)), dim3(256), 0, st, g, tail, m);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_axpy(double alpha, const double* x, double* y, int64_t len, hipStream_t st) {
    if (len == 0) return;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, alpha, x, y, len);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_fill(double value, double* y, int64_t len, hipStream_t st) {
    if (len == 0) return;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, value, y, len);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_gather_assemble(int64_t nnz, const int32_t* cptr, const int32_t* cidx, const double* slab,
                            double* Hval, bool long_lists, hipStream_t st, int32_t chunk, int32_t nchunk, double* part,
                            const int32_t* qmap, int64_t nq) {
    if (nnz == 0) return;
    if (!qmap) nq = nnz;
    if (nq == 0) return;
    if (long_lists && nchunk > 1) {
        hipLaunchKernelGGL(gather_assemble_chunk_kernel, dim3((unsigned)((nq * nchunk + 3) / 4)), dim3(256), 0, st, nq, qmap, chunk,
                           nchunk, cptr, cidx, slab, part);
        hipLaunchKernelGGL(gather_assemble_chunk_reduce, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, nq, qmap, nchunk, part, Hval);
    } else if (long_lists)
        hipLaunchKernelGGL(gather_assemble_wave_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, nq, qmap, cptr,
                           cidx, slab, Hval);
    else
        hipLaunchKernelGGL(gather_assemble_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, nq, qmap, cptr,
                           cidx, slab, Hval);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_gather_shared(int64_t nshared, const int32_t* sh_q, const int32_t* cptr, const int32_t* cidx, const double* slab,
                          double* out, hipStream_t st) {
    if (nshared == 0) return;
    hipLaunchKernelGGL(gather_shared_kernel, dim3((unsigned)((nshared + 255) / 256)), dim3(256), 0, st, nshared, sh_q, cptr, cidx,
                       slab, out);
    MGB_HIP_CHECK(hipGetLastError());
}

size_t panel_accumulate_lds(int p, int nu, int ctmax) {      // staging of one element, four waves (slab variant)
    const size_t per_wave = (size_t)p * ctmax + (size_t)(nu * (nu + 1) / 2) * p * p + (size_t)nu * p * ctmax + ctmax;
    return 4 * per_wave * sizeof(double);
}

bool panel_accumulate_fits(int p, int nu, int ctmax) {        // register staging limits of panel_accumulate_kernel
    return p * ctmax <= 4 * 256 && (nu * (nu + 1) / 2) * p * p <= 3 * 256 && ctmax <= 256;
}

size_t panel_stage_doubles(int p, int nu, int ctmax) {        // staging of one element, one workgroup
    return (size_t)p * ctmax + (size_t)(nu * (nu + 1) / 2) * p * p + (size_t)nu * p * ctmax + ctmax;
}

void launch_panel_accumulate(const PanelParams& P, const int32_t* ecols, int32_t m, int32_t nstream, int32_t nsplit,
                             int32_t chunk, int32_t ctmax, double* partial, double* H, hipStream_t st) {
    if (m == 0) return;
    const size_t lds = ((size_t)chunk + panel_stage_doubles(P.p, P.nu, ctmax)) * sizeof(double);
    MGB_REQUIRE(lds <= PANEL_ACC_LDS_MAX, "coarse-level accumulator + panels exceed the LDS budget");
    MGB_REQUIRE(panel_accumulate_fits(P.p, P.nu, ctmax), "coarse-level panels exceed the register staging of the accumulation kernel");
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)panel_accumulate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)PANEL_ACC_LDS_MAX);
        (void)hipGetLastError();
    });
    hipLaunchKernelGGL(panel_accumulate_kernel, dim3((unsigned)nstream, (unsigned)nsplit), dim3(256), lds, st, P, ecols, m,
                       ctmax, chunk, partial);
    hipLaunchKernelGGL(accumulate_reduce_kernel, dim3((unsigned)(((int64_t)m * m + 255) / 256)), dim3(256), 0, st, m,
                       nstream, partial, H);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_panel_project_staged(const PanelParams& P, int32_t ctmax, hipStream_t st) {
    if (P.N == 0) return;
    const size_t lds = panel_accumulate_lds(P.p, P.nu, ctmax);
    MGB_REQUIRE(lds <= PANEL_ACC_LDS_MAX, "coarse-level panels too wide for the staged projection kernel");
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)panel_project_staged_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)PANEL_ACC_LDS_MAX);
        (void)hipGetLastError();
    });
    hipLaunchKernelGGL(panel_project_staged_kernel, dim3((unsigned)((P.N + 3) / 4)), dim3(256), lds, st, P, ctmax);
    MGB_HIP_CHECK(hipGetLastError());
}

bool launch_panel_project_mfma(const PanelParams& P, hipStream_t st) {
    if (P.N == 0) return true;
    if (P.nu > MGBHIP_MAX_NU || P.p > 64) return false;
    const int ctpad = P.nu * pp_round_up(P.cmax, 16);            // every state's range padded to a tile
    const PanelMfmaLayout Y = panel_mfma_layout(P.p, P.nu, ctpad);
    const size_t lds = (size_t)Y.total * sizeof(double);
    static const bool off = [] { const char* e = getenv("MGBHIP_NO_MFMA_PROJECT"); return e && e[0] == '1'; }();
    // narrow supports (2-D hierarchies: a dozen columns per element) are faster through the staged loop kernel, four elements
    // per workgroup (L = 9: 120 us per launch); wide ones (3-D) are not
    if (off || lds > 64 * 1024 || ctpad < 48) return false;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)panel_project_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipGetLastError();
    });
    hipLaunchKernelGGL(panel_project_mfma_kernel, dim3((unsigned)P.N), dim3(256), lds, st, P, ctpad);
    MGB_HIP_CHECK(hipGetLastError());
    return true;
}

void launch_invert_lists(const int32_t* cidx, int64_t total, int32_t* spos, hipStream_t st) {
    if (total <= 0) return;
    hipLaunchKernelGGL(invert_lists_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, cidx, total, spos);
    MGB_HIP_CHECK(hipGetLastError());
}

void launch_panel_project(const PanelParams& P, hipStream_t st) {
    if (P.N == 0) return;
    const size_t lds = (size_t)4 * P.p * P.cmax * sizeof(double);
    // 64-node elements with full-width panels (fem3d k = 3 on a geometric ladder) need 128 KB: opt in once
    static const bool big_lds = hipFuncSetAttribute((const void*)panel_project_kernel,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) == hipSuccess;
    if (!big_lds) (void)hipGetLastError();
    MGB_REQUIRE(lds <= (big_lds ? 144 : 64) * 1024, "coarse-level panels too wide for the projection kernel");
    hipLaunchKernelGGL(panel_project_kernel, dim3((unsigned)((P.N + 3) / 4)), dim3(256), lds, st, P);
    MGB_HIP_CHECK(hipGetLastError());
}

}  // namespace mgbhip
