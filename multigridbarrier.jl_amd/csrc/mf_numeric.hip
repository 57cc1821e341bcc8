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

#include <cstdio>

#include <algorithm>
#include <cstdlib>

#include "../../include/mgbhip.h"
#include "mf_solver.hpp"

namespace mgbhip {

namespace {

#ifdef MGB_STEP_PROBE      // development probe build only (tools/gpu_probe.py)
__device__ long long g_probe[64];
#define SPL(i) do { if (threadIdx.x == 0 && gridDim.x == 1024 && blockIdx.x == 700 && j0 == 0) g_probe[48 + i] = wall_clock64(); } while (0)
#define SP(i) do { if (threadIdx.x == 0 && gridDim.x == 4096 && blockIdx.x == 3000) g_probe[40 + i] = wall_clock64(); if (threadIdx.x == 0 && gridDim.x == 1024 && blockIdx.x == 700) g_probe[24 + i] = wall_clock64(); } while (0)
#else
#define SPL(i) do { } while (0)
#define SP(i) do { } while (0)
#endif

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int TX = 16;    // row lanes of the 2-D thread maps
constexpr int NB = 32;    // panel width of the large-front path
constexpr int CT = 8;     // destination columns per workgroup in the large-front assembly (2 per wave)
constexpr int TR = 256;   // rows per workgroup in the panel solve
constexpr int ST = 64;    // tile edge of the symmetric update
constexpr int ASM_REL_LDS = 2048; // relative indices of one child kept in LDS by the large-front assembly
constexpr int CHILD_CHUNK = 64;   // child descriptors staged in LDS at a time

// ------------------------------------------------------------------------------------------------
// small fronts
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Broadcast of a double from a compile-time lane through SGPRs (v_readlane_b32 x 2): cheaper than
// the LDS-crossbar path of __shfl when the source lane is a constant after unrolling.
__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), srclane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// Exchange of a double inside a quad of lanes on the data-parallel path (two v_mov_b32 dpp): quad_perm control
// 0xB1 = lanes [1,0,3,2] (xor 1), 0x4E = [2,3,0,1] (xor 2).  __shfl_xor goes through ds_bpermute, i.e. the LDS pipeline.
template <int CTRL>
__device__ __forceinline__ double quad_perm_f64(double v) {      // also row_ror:n (0x120 + n): rotation inside 16 lanes
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// Sum over a row of 16 lanes (every lane ends with it): no LDS traffic.
__device__ __forceinline__ double row16_sum_f64(double v) {
    v += quad_perm_f64<0xB1>(v);
    v += quad_perm_f64<0x4E>(v);
    v += quad_perm_f64<0x124>(v);
    v += quad_perm_f64<0x128>(v);
    return v;
}
// Sum over the wave: rows on the data-parallel path, the four rows through the crossbar (2 exchanges instead of 6).
__device__ __forceinline__ double wave_sum_f64(double v) {
    v = row16_sum_f64(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// 1/d by v_rcp_f64 and two Newton steps (the pivot chain of the in-register LDL' is latency
// bound; the full IEEE division sequence is twice as long).  Error < 1 ulp of the quotient, and
// LDL' is backward stable under any such perturbation of the multipliers.
__device__ __forceinline__ double fast_recip(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, x, 1.0);
    x = __builtin_fma(x, e, x);
    e = __builtin_fma(-d, x, 1.0);
    x = __builtin_fma(x, e, x);
    return x;
}

// Entry (r, j), r >= j, of a child's update block.  Square children: U = offset of (k, k), M = the child's m (> 0).
// Packed leaf children (FrontDev::packed): U = offset of (k, k) in the packed triangle, M = -(m - k).
__device__ __forceinline__ int64_t child_entry(int64_t U, int32_t M, int j, int r) {
    return M > 0 ? U + (int64_t)j * M + r : U + (int64_t)j * (-M) - (j * (j - 1)) / 2 + (r - j);
}
__device__ __forceinline__ void child_update_desc(const FrontDev& C, int64_t& U, int32_t& M) {
    if (C.packed) {
        U = C.F_off + (int64_t)C.k * C.m - (C.k * (C.k - 1)) / 2;
        M = -(C.m - C.k);
    } else {
        U = C.F_off + (int64_t)C.k * C.m + C.k;
        M = C.m;
    }
}
__device__ __forceinline__ int64_t tiny_entry(const FrontDev& F, int r, int c) {        // (r, c), r >= c, of a leaf front
    return F.packed ? (int64_t)c * F.m - (c * (c - 1)) / 2 + (r - c) : r + (int64_t)c * F.m;
}

// In-register LDL' of an nb x nb block: lane r holds row r of the lower triangle in a[0..r].
// 32 x 31 / 2 shuffle + FMA pairs, no memory traffic; nb is wave-uniform.
template <int NBT>
__device__ __forceinline__ bool wave_ldlt_regs(double (&a)[NBT], int nb, int lane) {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
        if (j < nb) {
            const double d = readlane_f64(a[j], j);
            if (d == 0.0 || !isfinite(d)) bad = true;
            const double inv = fast_recip(d);
            const double aj = a[j];          // this lane's unscaled entry of column j
            const double lr = aj * inv;
#pragma unroll
            for (int c = j + 1; c < NBT; ++c) {
                // unscaled entry (c, j).  No lane predicate: lanes above the diagonal (lane < c)
                // only touch their never-read upper-triangle slots, and rows/columns >= nb hold
                // zeros, so the update is a plain FMA with an SGPR operand.
                const double v = readlane_f64(aj, c);
                a[c] -= lr * v;
            }
            if (lane > j) a[j] = lr;
        }
    }
    return bad;
}

// One workgroup per front.  Right-looking LDL' blocked by NB = 32 columns: wave 0 factors the
// diagonal block in registers (shuffles only), every thread then solves one panel row in
// registers, and all threads apply the rank-32 update -- 3 workgroup barriers per 32 columns.
template <int NBT, bool PACKED>
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
    // PACKED (classes 88 and 128, whose square arrays leave room for one or two workgroups per compute unit): the
    // front lives in LDS as a packed lower triangle, column c at W + co(c) (entry (r, c), r >= c, at co(c) + r) --
    // half the LDS, twice the resident fronts; analyze() remaps a_dst.  The smaller classes keep the square array
    // (their occupancy is not LDS-bound and the plain column offset c * m is cheaper to form).
    const int mm = PACKED ? m * (m + 1) / 2 : m * m;
    auto co = [m](int c) { return PACKED ? c * (m - 1) - c * (c - 1) / 2 : c * m; };
    double* S = W + mm;                        // [NBT][m] scaled multipliers of the current panel

    SP(0);
    // the first batch of A entries and the first chunk of child descriptors are requested before the LDS front is
    // zeroed: two dependent-load chains (a_src -> Hval, children -> fr) run under the fill instead of after it
    int a_d0 = -1;
    double a_v0 = 0.0;
    if (tid < F.a_cnt) {
        a_d0 = a_dst[F.a_off + tid];
        a_v0 = Hval[a_src[F.a_off + tid]];
    }
    int64_t pU = 0, pR = 0;
    int32_t pM = 0, pB = 0;
    if (tid < min(CHILD_CHUNK, F.nchild)) {
        const FrontDev C = fr[children[F.child_off + tid]];
        child_update_desc(C, pU, pM);
        pR = C.rel_off;
        pB = C.m - C.k;
    }
    for (int i = tid; i < mm; i += nt) W[i] = 0.0;
    __syncthreads();
    if (a_d0 >= 0) W[a_d0] = a_v0;
    for (int t = tid + nt; t < F.a_cnt; t += nt) W[a_dst[F.a_off + t]] = Hval[a_src[F.a_off + t]];
    __syncthreads();
    SP(1);
    // Extend-add of the children.  The additions of different children may hit the same slot, so
    // children stay ordered (deterministic sums) with a barrier between them -- but their global
    // loads do not have to: the child descriptors are fetched once into LDS, and the entries of
    // four children at a time are staged in registers before the first of them is applied, so a
    // front with many small children (static-condensation leaves under an element patch) pays one
    // memory latency per four children instead of three dependent ones per child.
    __shared__ int64_t cU[CHILD_CHUNK];        // arena offset of the child's update block (kc, kc)
    __shared__ int64_t cR[CHILD_CHUNK];        // rel offset
    __shared__ int32_t cM[CHILD_CHUNK], cB[CHILD_CHUNK];
    __shared__ int32_t crl[128];               // relative indices of one larger child (b <= m <= 128)
    __shared__ double prinv[32];               // reciprocal pivots of the current panel
    for (int cbase = 0; cbase < F.nchild; cbase += CHILD_CHUNK) {
        const int nc = min(CHILD_CHUNK, F.nchild - cbase);
        __syncthreads();
        if (tid < nc) {
            if (cbase == 0) {
                cU[tid] = pU; cR[tid] = pR; cM[tid] = pM; cB[tid] = pB;
            } else {
                const FrontDev C = fr[children[F.child_off + cbase + tid]];
                int64_t u_; int32_t m_;
                child_update_desc(C, u_, m_);
                cU[tid] = u_;
                cR[tid] = C.rel_off;
                cM[tid] = m_;
                cB[tid] = C.m - C.k;
            }
        }
        __syncthreads();
        for (int c0 = 0; c0 < nc; c0 += 4) {
            if (nt == 256 && (c0 & 15) == 0) {
                // Sixteen small children (update block <= 8 x 8) at once: wave w takes children 4w .. 4w+3, one entry
                // per lane, so ALL their loads are in flight together (one memory latency for the group instead
                // of four).  The additions keep child order: a wave applies its four children in program order
                // (LDS operations of one wave stay ordered) and the waves take turns, four barriers in all.
                const int ng = min(16, nc - c0);
                bool small16 = true;
                for (int u = 0; u < ng; ++u) small16 = small16 && cB[c0 + u] * cB[c0 + u] <= 64;
                if (small16) {
                    const int w = tid >> 6, e = tid & 63;
                    int dst[4];
                    double val[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        dst[u] = -1;
                        val[u] = 0.0;
                        const int c = c0 + 4 * w + u;
                        if (4 * w + u < ng) {
                            const int b = cB[c];
                            if (e < b * b) {
                                const int j = e / b, r = e - j * b;
                                if (r >= j) {
                                    const int32_t* rl = rel + cR[c];
                                    dst[u] = rl[r] + co(rl[j]);
                                    val[u] = arena[child_entry(cU[c], cM[c], j, r)];
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph) {
                        if (4 * ph < ng) {
                            if (ph == w) {
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    if (dst[u] >= 0) W[dst[u]] += val[u];
                                    wave_sync();            // child u's stores before child u+1's loads
                                }
                            }
                            __syncthreads();
                        }
                    }
                    c0 += 12;          // the loop increment adds the other 4
                    continue;
                }
            }
            int dst[4];
            double val[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dst[u] = -1;
                val[u] = 0.0;
                const int c = c0 + u;
                if (c < nc) {
                    const int b = cB[c];
                    if (b * b <= nt && tid < b * b) {
                        const int j = tid / b, r = tid - j * b;
                        if (r >= j) {
                            const int32_t* rl = rel + cR[c];
                            dst[u] = rl[r] + co(rl[j]);
                            val[u] = arena[child_entry(cU[c], cM[c], j, r)];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u;
                if (c >= nc) break;
                if (dst[u] >= 0) W[dst[u]] += val[u];
                const int b = cB[c];
                if (b * b > nt) {               // larger child: 2-D sweep, relative indices from LDS
                    const int32_t* rlg = rel + cR[c];
                    for (int j = tid; j < b; j += nt) crl[j] = rlg[j];
                    __syncthreads();
                    const int mc = cM[c];
                    for (int j = ty; j < b; j += TYn) {
                        const int dcol = co(crl[j]);
                        const double* Uc = arena + child_entry(cU[c], mc, j, 0);     // (r, j) at Uc[r]; packed children shift by j
                        for (int r = j + tx; r < b; r += 4 * TX) {      // four rows per lane in flight
                            const int r1 = r + TX, r2 = r + 2 * TX, r3 = r + 3 * TX;
                            const double u0 = Uc[r];
                            const double u1 = r1 < b ? Uc[r1] : 0.0;
                            const double u2 = r2 < b ? Uc[r2] : 0.0;
                            const double u3 = r3 < b ? Uc[r3] : 0.0;
                            W[crl[r] + dcol] += u0;
                            if (r1 < b) W[crl[r1] + dcol] += u1;
                            if (r2 < b) W[crl[r2] + dcol] += u2;
                            if (r3 < b) W[crl[r3] + dcol] += u3;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    SP(2);
    bool bad = false;
    for (int j0 = 0; j0 < k; j0 += NBT) {
        const int nb = min(NBT, k - j0);
        SPL(0);
        if (tid < 64) {                       // diagonal block in registers
            double a[NBT];
#pragma unroll
            for (int c = 0; c < NBT; ++c) a[c] = (tid < nb && c <= tid) ? W[(j0 + tid) + co(j0 + c)] : 0.0;
            bad |= wave_ldlt_regs<NBT>(a, nb, tid);
#pragma unroll
            for (int c = 0; c < NBT; ++c)
                if (tid < nb && c <= tid) {
                    W[(j0 + tid) + co(j0 + c)] = a[c];
                    if (c == tid) prinv[c] = 1.0 / a[c];      // pivot reciprocals for the row solves
                }
        }
        SPL(1);
        __syncthreads();
        SPL(2);
        {                                     // panel rows: l = (a L11^{-T}) D^{-1}, one row per thread
            const int r = j0 + nb + tid;
            if (r < m) {
                double a[NBT];
#pragma unroll
                for (int c = 0; c < NBT; ++c) a[c] = (c < nb) ? W[r + co(j0 + c)] : 0.0;
#pragma unroll
                for (int c = 0; c < NBT; ++c) {
                    if (c < nb) {
                        double v = a[c];
#pragma unroll
                        for (int q = 0; q < NBT; ++q)
                            if (q < c) v -= a[q] * W[(j0 + c) + co(j0 + q)];
                        a[c] = v;
                    }
                }
                // a[c] is still l(r, c) * d_c here: exactly the scaled multiplier the update needs
#pragma unroll
                for (int c = 0; c < NBT; ++c) {
                    S[c * m + r] = (c < nb) ? a[c] : 0.0;
                    if (c < nb) W[r + co(j0 + c)] = a[c] * prinv[c];
                }
            }
        }
        SPL(3);
        __syncthreads();
        SPL(4);
        // rank-nb update of the trailing lower triangle on the matrix cores, one 16 x 16 tile per wave and pass:
        //   W[r, c] -= sum_q l(r, q) * S(q, c),   S(q, c) = l(c, q) d_q from the row solve above.
        // v_mfma_f64_16x16x4: lane (fr16, fk) feeds A[m = fr16][k = fk] = S(q, cc0 + fr16) and B[k = fk][n = fr16] =
        // l(r0 + fr16, q), and holds D[m = fk + 4 i][n = fr16], i = 0..3 -- rows run along the 16 lanes, so every
        // LDS access of a tile is 16 consecutive doubles.  The scalar form read 16 LDS operands per 2 FMAs and kept
        // the LDS pipeline of the compute unit saturated (three fronts per unit: 17 us for a rank-15 update of 65 rows).
        {
            const int c0 = j0 + nb;
            const int T = (m - c0 + 15) >> 4;
            const int lane = tid & 63, fr16 = lane & 15, fk = lane >> 4;
            for (int tile = tid >> 6; tile < T * (T + 1) / 2; tile += nt >> 6) {
                int I = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
                while ((I + 1) * (I + 2) / 2 <= tile) ++I;
                while (I * (I + 1) / 2 > tile) --I;
                const int J = tile - I * (I + 1) / 2;
                const int r0 = c0 + 16 * I, cc0 = c0 + 16 * J;
                const int rr = min(r0 + fr16, m - 1), cc = min(cc0 + fr16, m - 1);       // tiles overhang the front: clamp, never stored
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < NBT / 4; ++kk) {
                    const int q = 4 * kk + fk;
                    const double sa = S[q * m + cc];                                   // rows q >= nb of S are zero
                    const double lb = (q < nb) ? W[rr + co(j0 + q)] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sa, lb, acc, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = cc0 + fk + 4 * i, row = r0 + fr16;
                    if (row < m && col < m && row >= col) W[row + co(col)] -= acc[i];
                }
            }
        }
        SPL(5);
        __syncthreads();
        SPL(6);
    }
    SP(3);
    if (bad && tid == 0) atomicOr(status, 1);
    // write the lower triangle back to the (square, column-major) frontal matrix: a 2-D sweep, rows fastest
    if (PACKED) {
        for (int c = ty; c < m; c += TYn) {
            const int cc = co(c);
            for (int r = c + tx; r < m; r += TX) Fg[r + (int64_t)c * m] = W[cc + r];
        }
    } else {
        for (int i = tid; i < mm; i += nt) Fg[i] = W[i];
    }
    SP(4);
}

// Fronts with m <= MW (32 or 48): ONE WAVE per front, no workgroup barriers.  The front is assembled in a packed
// LDS triangle (column stride MW; analyze() remaps a_dst for these fronts), lane r then takes row r into
// registers and the whole partial factorization -- k pivots and the Schur complement of the boundary rows -- is
// wave_ldlt_regs: v_readlane broadcasts and FMAs only.  A 46-row front with 21 pivots and 16 leaf children takes
// about half the time of the workgroup-per-front kernel, and twice as many fronts are resident per compute unit.
template <int MW>
__global__ __launch_bounds__(256) void mf_factor_wave(const FrontDev* __restrict__ fr, int32_t first, int32_t count,
                                                      const int32_t* __restrict__ children, const int32_t* __restrict__ rel,
                                                      const int32_t* __restrict__ a_src, const int32_t* __restrict__ a_dst,
                                                      const double* __restrict__ Hval, double* __restrict__ arena,
                                                      int32_t* __restrict__ status) {
    constexpr int PK = MW * (MW + 1) / 2;
    extern __shared__ double sh[];
#ifdef MGB_STEP_PROBE      // one wave of the level-1 launch at L = 9: phase timestamps (tools/gpu_probe_wave.py)
#define WP(i) do { if (threadIdx.x == 0 && gridDim.x == 2048 && blockIdx.x == 1500) g_probe[8 + (i)] = wall_clock64(); } while (0)
#else
#define WP(i) do { } while (0)
#endif
    WP(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = blockIdx.x * 4 + wave;
    if (fi >= count) return;                       // waves are independent: no workgroup barrier below
    const FrontDev F = fr[first + fi];
    const int m = F.m, k = F.k;
    double* W = sh + (size_t)wave * PK;
    // child descriptors of this wave (chunks of WCH), behind the four fronts: 4 x 48-row triangles + the descriptors
    // stay under 40 KB, so four workgroups share a compute unit
    constexpr int WCH = 32;
    int64_t* cU = reinterpret_cast<int64_t*>(sh + (size_t)4 * PK) + wave * 2 * WCH;
    int64_t* cR = cU + WCH;
    int32_t* cM = reinterpret_cast<int32_t*>(reinterpret_cast<int64_t*>(sh + (size_t)4 * PK) + 4 * 2 * WCH) + wave * 2 * WCH;
    int32_t* cB = cM + WCH;
    auto pidx = [](int r, int c) { return c * MW - c * (c - 1) / 2 + (r - c); };
    // first batch of A entries and the first chunk of child descriptors are requested before the triangle is zeroed
    int a_d0 = -1;
    double a_v0 = 0.0;
    if (lane < F.a_cnt) {
        a_d0 = a_dst[F.a_off + lane];
        a_v0 = Hval[a_src[F.a_off + lane]];
    }
    int64_t pU = 0, pR = 0;
    int32_t pM = 0, pB = 0;
    if (lane < min(WCH, F.nchild)) {
        const FrontDev C = fr[children[F.child_off + lane]];
        child_update_desc(C, pU, pM);
        pR = C.rel_off;
        pB = C.m - C.k;
    }
    for (int i = lane; i < PK; i += 64) W[i] = 0.0;
    wave_sync();
    WP(1);
    if (a_d0 >= 0) W[a_d0] = a_v0;
    for (int t = lane + 64; t < F.a_cnt; t += 64) W[a_dst[F.a_off + t]] = Hval[a_src[F.a_off + t]];
    wave_sync();
    WP(2);
    for (int cbase = 0; cbase < F.nchild; cbase += WCH) {
        const int nc = min(WCH, F.nchild - cbase);
        if (lane < nc) {
            if (cbase == 0) {
                cU[lane] = pU; cR[lane] = pR; cM[lane] = pM; cB[lane] = pB;
            } else {
                const FrontDev C = fr[children[F.child_off + cbase + lane]];
                int64_t u_; int32_t m_;
                child_update_desc(C, u_, m_);
                cU[lane] = u_;
                cR[lane] = C.rel_off;
                cM[lane] = m_;
                cB[lane] = C.m - C.k;
            }
        }
        wave_sync();
        for (int c0 = 0; c0 < nc; c0 += 16) {
            const int ng = min(16, nc - c0);
            bool small16 = true;
            for (int u = 0; u < ng; ++u) small16 = small16 && cB[c0 + u] * cB[c0 + u] <= 64;
            if (small16) {
                // sixteen small children (the element leaves under a level-1 front): one entry per lane and child, all
                // loads in flight at once, then added in child order.  j = lane / b by a float reciprocal: exact for
                // lane < 64, b <= 8 ((lane + 1/2) / b stays 1/16 away from every integer).
                int dst[16];
                double val[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    dst[u] = -1;
                    val[u] = 0.0;
                    if (u < ng) {
                        const int c = c0 + u, b = cB[c];
                        if (lane < b * b) {
                            const int j = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)b)), r = lane - j * b;
                            if (r >= j) {
                                const int32_t* rl = rel + cR[c];
                                dst[u] = pidx(rl[r], rl[j]);
                                val[u] = arena[child_entry(cU[c], cM[c], j, r)];
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (u < ng) {
                        if (dst[u] >= 0) W[dst[u]] += val[u];
                        wave_sync();                    // child u's stores before child u+1's loads
                    }
                }
            } else {
                for (int u = 0; u < ng; ++u) {
                    const int c = c0 + u, b = cB[c], mc = cM[c];
                    const int32_t* rl = rel + cR[c];
                    for (int e = lane; e < b * b; e += 256) {       // four entries per lane in flight
                        int dd[4];
                        double vv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int ee = e + 64 * q;
                            dd[q] = -1;
                            vv[q] = 0.0;
                            if (ee < b * b) {
                                const int j = ee / b, r = ee - j * b;
                                if (r >= j) {
                                    dd[q] = pidx(rl[r], rl[j]);
                                    vv[q] = arena[child_entry(cU[c], mc, j, r)];
                                }
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (dd[q] >= 0) W[dd[q]] += vv[q];         // distinct slots within one child
                    }
                    wave_sync();
                }
            }
        }
        wave_sync();
    }
    WP(3);
    double a[MW];
#pragma unroll
    for (int c = 0; c < MW; ++c) a[c] = (lane < m && c <= lane) ? W[pidx(lane, c)] : 0.0;
    WP(4);
    const bool bad = wave_ldlt_regs<MW>(a, k, lane);
    WP(5);
    if (bad) atomicOr(status, 1);
    if (lane < m) {
        double* Fg = arena + F.F_off;
#pragma unroll
        for (int c = 0; c < MW; ++c)
            if (c <= lane) Fg[lane + (int64_t)c * m] = a[c];
    }
    WP(6);
}

// Triangular solves of small fronts: one wave per front (4 fronts per workgroup), the work
// vector lives in registers (rows lane and lane + 64), no workgroup barriers.
__global__ __launch_bounds__(256) void mf_forward_small(const FrontDev* __restrict__ fr, int32_t first,
                                                        int32_t count, int32_t ts,
                                                        const int32_t* __restrict__ front_idx,
                                                        const int32_t* __restrict__ children,
                                                        const int32_t* __restrict__ rel,
                                                        const double* __restrict__ arena,
                                                        const double* __restrict__ b, double* __restrict__ y,
                                                        double* __restrict__ uvec) {
    extern __shared__ double sh[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = blockIdx.x * 4 + wave;
    if (fi >= count) return;
    const FrontDev F = fr[first + fi];
    const int m = F.m, k = F.k;
    double* t = sh + wave * ts;
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    for (int j = lane; j < m; j += 64) t[j] = (j < k) ? b[idx[j]] : 0.0;
    wave_sync();
    // children's update vectors: descriptors of up to 64 children are fetched by the lanes in
    // parallel and broadcast from registers; the loads of four children are in flight together,
    // the additions stay in child order (deterministic sums)
    for (int cbase = 0; cbase < F.nchild; cbase += 64) {
        const int nc = min(64, F.nchild - cbase);
        int64_t my_u = 0, my_r = 0;
        int my_b = 0;
        if (lane < nc) {
            const FrontDev C = fr[children[F.child_off + cbase + lane]];
            my_u = C.u_off;
            my_r = C.rel_off;
            my_b = C.m - C.k;
        }
        for (int c0 = 0; c0 < nc; c0 += 4) {
            int dst[4];
            double val[4];
            int bb[4];
            int64_t ru[4], rr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = min(c0 + u, nc - 1);
                bb[u] = (c0 + u < nc) ? __shfl(my_b, c, 64) : 0;
                ru[u] = __shfl(my_u, c, 64);
                rr[u] = __shfl(my_r, c, 64);
                dst[u] = -1;
                val[u] = 0.0;
                if (lane < bb[u]) {
                    dst[u] = rel[rr[u] + lane];
                    val[u] = uvec[ru[u] + lane];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c0 + u >= nc) break;
                if (dst[u] >= 0) t[dst[u]] += val[u];
                for (int j = lane + 64; j < bb[u]; j += 64) t[rel[rr[u] + j]] += uvec[ru[u] + j];
                wave_sync();
            }
        }
    }
    const int r1 = lane + 64;
    double t0 = (lane < m) ? t[lane] : 0.0;
    double t1 = (r1 < m) ? t[r1] : 0.0;
#pragma unroll 4
    for (int j = 0; j < k; ++j) {
        const double* Lj = Fm + (int64_t)j * m;
        const double l0 = (lane > j && lane < m) ? Lj[lane] : 0.0;
        const double l1 = (r1 > j && r1 < m) ? Lj[r1] : 0.0;
        const double tj = (j < 64) ? readlane_f64(t0, j) : readlane_f64(t1, j - 64);   // j is wave-uniform
        t0 -= l0 * tj;
        t1 -= l1 * tj;
    }
    if (lane < m) {
        if (lane < k) y[idx[lane]] = t0 / Fm[lane + (int64_t)lane * m];
        else uvec[F.u_off + lane - k] = t0;
    }
    if (r1 < m) {
        if (r1 < k) y[idx[r1]] = t1 / Fm[r1 + (int64_t)r1 * m];
        else uvec[F.u_off + r1 - k] = t1;
    }
}

// KMAX > 0: every front of the launch has k <= KMAX pivots and the lane's entries of all pivot columns are requested
// before the first elimination step (the steps are a dependent chain; with the loads inside it every step paid a
// memory latency).  KMAX == 0: the rolled form.
template <int KMAX>
__global__ __launch_bounds__(256) void mf_backward_small(const FrontDev* __restrict__ fr, int32_t first,
                                                         int32_t count,
                                                         const int32_t* __restrict__ front_idx,
                                                         const double* __restrict__ arena,
                                                         const double* __restrict__ y, double* __restrict__ x) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = blockIdx.x * 4 + wave;
    if (fi >= count) return;
    const FrontDev F = fr[first + fi];
    const int m = F.m, k = F.k;
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    const int r1 = lane + 64;
    double t0 = 0.0, t1 = 0.0;
    if constexpr (KMAX > 0) {
        double l0[KMAX], l1[KMAX];
        const bool two = m > 64;                     // wave-uniform
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            l0[j] = (j < k && lane > j && lane < m) ? Fm[(int64_t)j * m + lane] : 0.0;
            l1[j] = (two && j < k && r1 < m) ? Fm[(int64_t)j * m + r1] : 0.0;        // r1 > j always (k <= KMAX <= 64)
        }
        if (lane < m) t0 = (lane < k) ? y[idx[lane]] : x[idx[lane]];
        if (r1 < m) t1 = (r1 < k) ? y[idx[r1]] : x[idx[r1]];
#pragma unroll
        for (int j = KMAX - 1; j >= 0; --j) {
            if (j < k) {
                double s = l0[j] * t0;
                if (two) s += l1[j] * t1;
                s = wave_sum_f64(s);
                if (lane == j) t0 -= s;
            }
        }
    } else {
        if (lane < m) t0 = (lane < k) ? y[idx[lane]] : x[idx[lane]];
        if (r1 < m) t1 = (r1 < k) ? y[idx[r1]] : x[idx[r1]];
#pragma unroll 2
        for (int j = k - 1; j >= 0; --j) {
            const double* Lj = Fm + (int64_t)j * m;
            double s = 0.0;
            if (lane > j && lane < m) s += Lj[lane] * t0;
            if (r1 > j && r1 < m) s += Lj[r1] * t1;
            s = wave_sum_f64(s);
            if (lane == j) t0 -= s;
            if (r1 == j) t1 -= s;
        }
    }
    if (lane < k) x[idx[lane]] = t0;
    if (r1 < k) x[idx[r1]] = t1;
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
                                                       const int32_t* __restrict__ a_colptr,
                                                       const double* __restrict__ Hval, double* __restrict__ arena) {
    __shared__ int32_t rls[ASM_REL_LDS];
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m;
    const int c0 = blockIdx.x * CT;
    if (c0 >= m) return;
    const int c1 = min(c0 + CT, m);
    double* W = arena + F.F_off;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;     // one wave per destination column, lanes on the rows
    for (int c = c0 + wave; c < c1; c += 4) {
        double* Wc = W + (int64_t)c * m;
        for (int r = c + lane; r < m; r += 64) Wc[r] = 0.0;
    }
    __syncthreads();
    {   // A entries are grouped by pivot column: the per-column offsets give the range of [c0, c1)
        const int32_t* cp = a_colptr + F.acol_off;
        const int beg = cp[min(c0, F.k)], end = cp[min(c1, F.k)];
        const int32_t* ad = a_dst + F.a_off;
        for (int t = beg + tid; t < end; t += 256) W[ad[t]] = Hval[a_src[F.a_off + t]];
    }
    __syncthreads();
    for (int c = 0; c < F.nchild; ++c) {
        const FrontDev C = fr[children[F.child_off + c]];
        const double* U = arena + C.F_off;
        const int mc = C.m, kc = C.k, b = mc - kc;
        const int32_t* rlg = rel + C.rel_off;
        // relative indices of this child in LDS: the two searches and the scatter below read them
        // from there instead of chasing ~2 log2(b) dependent global loads
        const bool in_lds = b <= ASM_REL_LDS;
        if (in_lds)
            for (int j = tid; j < b; j += 256) rls[j] = rlg[j];
        __syncthreads();
        const int32_t* rl = in_lds ? rls : rlg;
        // child columns whose destination lies in [c0, c1): rel is increasing
        int lo = 0, hi = b;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (rl[mid] < c0) lo = mid + 1; else hi = mid; }
        const int jb = lo;
        hi = b;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (rl[mid] < c1) lo = mid + 1; else hi = mid; }
        const int je = lo;
        for (int j = jb + wave; j < je; j += 4) {
            double* Wc = W + (int64_t)rl[j] * m;
            const double* Uc = U + (int64_t)(kc + j) * mc + kc;
            // the read-modify-write chain rl -> W is latency bound and W may alias U for the
            // compiler: stage four independent rows per lane so their loads are in flight together
            for (int r = j + lane; r < b; r += 256) {
                const int r1 = r + 64, r2 = r + 128, r3 = r + 192;
                const int i0 = rl[r];
                const int i1 = r1 < b ? rl[r1] : i0;
                const int i2 = r2 < b ? rl[r2] : i0;
                const int i3 = r3 < b ? rl[r3] : i0;
                const double u0 = Uc[r];
                const double u1 = r1 < b ? Uc[r1] : 0.0;
                const double u2 = r2 < b ? Uc[r2] : 0.0;
                const double u3 = r3 < b ? Uc[r3] : 0.0;
                const double w0 = Wc[i0], w1 = Wc[i1], w2 = Wc[i2], w3 = Wc[i3];
                Wc[i0] = w0 + u0;
                if (r1 < b) Wc[i1] = w1 + u1;
                if (r2 < b) Wc[i2] = w2 + u2;
                if (r3 < b) Wc[i3] = w3 + u3;
            }
        }
        __syncthreads();
    }
}

// Panel step.  Every workgroup of a front loads the (fully updated, still unfactored)
// diagonal block, wave 0 factors it redundantly in LDS, then the workgroup solves its TR
// rows of the panel: L21 = A21 L11^{-T} D^{-1}.  The factored diagonal block goes to a
// scratch slot (`dscr`), never in place, because sibling workgroups are still reading the
// unfactored block; the update kernel copies it home.  A front without rows below the block
// has a single active workgroup, which writes in place.
__global__ __launch_bounds__(256) void mf_big_panel(const FrontDev* __restrict__ fr, int32_t first, int j0,
                                                    double* __restrict__ arena, double* __restrict__ dscr,
                                                    int32_t* __restrict__ status, int do_diag) {
    __shared__ double Dk[NB][NB + 1];
    __shared__ double rinv[NB];
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m, k = F.k;
    if (j0 >= k) return;
    const int nb = min(NB, k - j0);
    const int r0 = j0 + nb + blockIdx.x * TR;
    if (blockIdx.x > 0 && r0 >= m) return;
    double* W = arena + F.F_off;
    double* slot = dscr + ((int64_t)blockIdx.y * 2 + ((j0 / NB) & 1)) * (NB * NB);
    const int tid = threadIdx.x;
    const bool last = (j0 + nb >= m);              // no panel rows, no trailing block: write home
    // this thread's panel row: issue the loads before the diagonal block is ready
    const int r = r0 + tid;
    double a[NB];
    if (r < m) {
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = (c < nb) ? W[r + (int64_t)(j0 + c) * m] : 0.0;
    }
    if (do_diag) {
        for (int i = tid; i < NB * NB; i += 256) {
            const int rr = i % NB, c = i / NB;
            Dk[rr][c] = (rr >= c && rr < nb) ? W[(j0 + rr) + (int64_t)(j0 + c) * m] : 0.0;
        }
        __syncthreads();
        if (tid < 64) {
            double d[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) d[c] = (tid < nb && c <= tid) ? Dk[tid][c] : 0.0;
            const bool bad = wave_ldlt_regs<NB>(d, nb, tid);
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if (tid < nb && c <= tid) Dk[tid][c] = d[c];
            if (bad && tid == 0 && blockIdx.x == 0) atomicOr(status, 1);
        }
        __syncthreads();
        if (blockIdx.x == 0 && !last) {
            for (int i = tid; i < NB * NB; i += 256) {
                const int rr = i % NB, c = i / NB;
                if (rr >= c && rr < nb) slot[rr + NB * c] = Dk[rr][c];
            }
        }
    } else {
        // factored by the previous step's update kernel (look-ahead)
        for (int i = tid; i < NB * NB; i += 256) {
            const int rr = i % NB, c = i / NB;
            Dk[rr][c] = (rr >= c && rr < nb) ? slot[rr + NB * c] : 0.0;
        }
        __syncthreads();
    }
    if (tid < nb) rinv[tid] = 1.0 / Dk[tid][tid];
    if (blockIdx.x == 0 && last) {
        for (int i = tid; i < NB * NB; i += 256) {
            const int rr = i % NB, c = i / NB;
            if (rr >= c && rr < nb) W[(j0 + rr) + (int64_t)(j0 + c) * m] = Dk[rr][c];
        }
    }
    __syncthreads();
    if (r < m) {
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
            if (c < nb) W[r + (int64_t)(j0 + c) * m] = a[c] * rinv[c];
    }
}

// Symmetric update of the trailing block with the finished panel:
// C[r, c] -= sum_q L[r, q] d_q L[c, q], 64 x 64 tiles of the lower triangle, 4 x 4 per thread.
// Tile 0 also copies the factored diagonal block from the scratch slot to its home.
__global__ __launch_bounds__(256) void mf_big_update(const FrontDev* __restrict__ fr, int32_t first, int j0,
                                                     double* __restrict__ arena, double* __restrict__ dscr,
                                                     int32_t* __restrict__ status) {
    __shared__ double Pi[NB][ST + 1];
    __shared__ double Qj[NB][ST + 1];
    __shared__ double dq[NB];
    __shared__ double Dn[NB][NB + 1];      // look-ahead: the next diagonal block (tile 0 only)
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m, k = F.k;
    if (j0 >= k) return;
    const int nb = min(NB, k - j0);
    const int j1 = j0 + nb;
    const int T = (m - j1 + ST - 1) / ST;
    const int tid = threadIdx.x;
    double* W = arena + F.F_off;
    const double* src = dscr + ((int64_t)blockIdx.y * 2 + ((j0 / NB) & 1)) * (NB * NB);
    const bool look = j1 < k;                  // a next panel exists: its diagonal block is factored here
    if (blockIdx.x == gridDim.x - 1) {
        // Look-ahead workgroup: update only the next diagonal block (nbn x nbn corner of tile 0)
        // and factor it, concurrently with the trailing tiles, so the next panel kernel starts
        // with its row solves at once.  Tile 0 leaves that corner alone (it is rewritten from
        // the scratch slot when the factored block goes home), so there is no race on W.
        if (!look) return;
        const int nbn = min(NB, k - j1);
        double* nslot = dscr + ((int64_t)blockIdx.y * 2 + ((j1 / NB) & 1)) * (NB * NB);
        if (tid < NB) dq[tid] = (tid < nb) ? src[tid + NB * tid] : 0.0;
        double w0[NB * NB / 256];                // corner entries, loaded while the panel rows arrive
#pragma unroll
        for (int t = 0; t < NB * NB / 256; ++t) {
            const int i = tid + 256 * t, rr = i % NB, c = i / NB;
            w0[t] = (rr >= c && rr < nbn) ? W[(j1 + rr) + (int64_t)(j1 + c) * m] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < NB * NB / 256; ++t) {
            const int i = tid + 256 * t, rr = i % NB, q = i / NB;
            Pi[q][rr] = (rr < nbn && q < nb) ? W[(j1 + rr) + (int64_t)(j0 + q) * m] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NB * NB / 256; ++t) {
            const int i = tid + 256 * t, rr = i % NB, c = i / NB;
            double acc = 0.0;
#pragma unroll 8
            for (int q = 0; q < NB; ++q) acc += Pi[q][rr] * (Pi[q][c] * dq[q]);
            Dn[rr][c] = w0[t] - acc;
        }
        __syncthreads();
        if (tid < 64) {
            double d[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) d[c] = (tid < nbn && c <= tid) ? Dn[tid][c] : 0.0;
            const bool bad = wave_ldlt_regs<NB>(d, nbn, tid);
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if (tid < nbn && c <= tid) nslot[tid + NB * c] = d[c];
            if (bad && tid == 0) atomicOr(status, 1);
        }
        return;
    }
    const int lin = blockIdx.x;
    int ti = (int)((sqrt(8.0 * lin + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= lin) ++ti;
    while (ti * (ti + 1) / 2 > lin) --ti;
    const int tj = lin - ti * (ti + 1) / 2;
    if (ti >= T) return;
    // One round of global loads: the pivots d_q, the two 64 x 32 panel slices and this thread's
    // 4 x 4 micro-tile of the trailing block are all requested before anything waits (the
    // scaling by d_q happens on the LDS side, the micro-tile is consumed after the products).
    if (tid < NB) dq[tid] = (tid < nb) ? src[tid + NB * tid] : 0.0;
    const int rbase = j1 + ti * ST, cbase = j1 + tj * ST;
    const int tx = tid % 16, ty = tid / 16;
    double wt[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int c = cbase + ty + 16 * b;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = rbase + tx + 16 * a;
            wt[a][b] = (c < m && r < m && r >= c) ? W[r + (int64_t)c * m] : 0.0;
        }
    }
    for (int i = tid; i < NB * ST; i += 256) {
        const int rr = i % ST, q = i / ST;
        const int r = rbase + rr, c = cbase + rr;
        Pi[q][rr] = (q < nb && r < m) ? W[r + (int64_t)(j0 + q) * m] : 0.0;
        Qj[q][rr] = (q < nb && c < m) ? W[c + (int64_t)(j0 + q) * m] : 0.0;
    }
    if (lin == 0) {
        for (int i = tid; i < nb * nb; i += 256) {
            const int r = i % nb, c = i / nb;
            if (r >= c) W[(j0 + r) + (int64_t)(j0 + c) * m] = src[r + NB * c];
        }
    }
    __syncthreads();
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll 4
    for (int q = 0; q < NB; ++q) {              // rows q >= nb of Pi/Qj and dq hold zeros
        double pr[4], qc[4];
        const double d = dq[q];
#pragma unroll
        for (int a = 0; a < 4; ++a) pr[a] = Pi[q][tx + 16 * a] * d;
#pragma unroll
        for (int b = 0; b < 4; ++b) qc[b] = Qj[q][ty + 16 * b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] += pr[a] * qc[b];
    }
    const int nskip = (lin == 0 && look) ? min(NB, k - j1) : 0;     // corner owned by the look-ahead workgroup
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int c = cbase + ty + 16 * b;
        if (c >= m) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = rbase + tx + 16 * a;
            if (r < m && r >= c && !(r - j1 < nskip && c - j1 < nskip)) W[r + (int64_t)c * m] = wt[a][b] - acc[a][b];
        }
    }
}

// ---- large-front triangular solves: multi-workgroup, one launch per 32-column block step ----
// Work vectors live in `tg` (indexed like front_idx); solved pivot blocks go to `ts` (forward)
// or straight to x (backward), never in place, because sibling workgroups still read them.

// t = [b(piv); 0] + children's update vectors; each workgroup owns 256 destination entries.
__global__ __launch_bounds__(256) void mf_fwd_big_init(const FrontDev* __restrict__ fr, int32_t first,
                                                       const int32_t* __restrict__ front_idx,
                                                       const int32_t* __restrict__ children,
                                                       const int32_t* __restrict__ rel,
                                                       const double* __restrict__ b,
                                                       const double* __restrict__ uvec, double* __restrict__ tg) {
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m, k = F.k;
    const int d0 = blockIdx.x * 256;
    if (d0 >= m) return;
    const int d1 = min(d0 + 256, m);
    const int tid = threadIdx.x;
    double* t = tg + F.idx_off;
    const int32_t* idx = front_idx + F.idx_off;
    const int jme = d0 + tid;
    double v = 0.0;
    if (jme < d1 && jme < k) v = b[idx[jme]];
    __shared__ double tl[256];
    tl[tid] = v;
    __syncthreads();
    for (int c = 0; c < F.nchild; ++c) {
        const FrontDev C = fr[children[F.child_off + c]];
        const int32_t* rl = rel + C.rel_off;
        const double* uc = uvec + C.u_off;
        const int bc = C.m - C.k;
        int lo = 0, hi = bc;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (rl[mid] < d0) lo = mid + 1; else hi = mid; }
        const int jb = lo;
        hi = bc;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (rl[mid] < d1) lo = mid + 1; else hi = mid; }
        for (int j = jb + tid; j < lo; j += 256) tl[rl[j] - d0] += uc[j];
        __syncthreads();
    }
    if (jme < d1) t[jme] = tl[tid];
}

__global__ __launch_bounds__(256) void mf_fwd_big_step(const FrontDev* __restrict__ fr, int32_t first, int j0,
                                                       const double* __restrict__ arena, double* __restrict__ tg,
                                                       double* __restrict__ ts) {
    __shared__ double Dk[NB][NB + 1];
    __shared__ double yb[NB];
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m, k = F.k;
    if (j0 >= k) return;
    const int nb = min(NB, k - j0);
    const int r0 = j0 + nb + blockIdx.x * 256;
    if (blockIdx.x > 0 && r0 >= m) return;
    const double* Fm = arena + F.F_off;
    double* t = tg + F.idx_off;
    const int tid = threadIdx.x;
    for (int i = tid; i < nb * nb; i += 256) {
        const int r = i % nb, c = i / nb;
        Dk[r][c] = (r > c) ? Fm[(j0 + r) + (int64_t)(j0 + c) * m] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        double v = (tid < nb) ? t[j0 + tid] : 0.0;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            if (c < nb) {
                const double tc = readlane_f64(v, c);
                if (tid > c && tid < nb) v -= Dk[tid][c] * tc;
            }
        }
        if (tid < nb) {
            yb[tid] = v;
            if (blockIdx.x == 0) ts[F.idx_off + j0 + tid] = v;
        }
    }
    __syncthreads();
    const int r = r0 + tid;
    if (r < m) {
        double v = t[r];
        for (int c = 0; c < nb; ++c) v -= Fm[r + (int64_t)(j0 + c) * m] * yb[c];
        t[r] = v;
    }
}

__global__ __launch_bounds__(256) void mf_fwd_big_fin(const FrontDev* __restrict__ fr, int32_t first,
                                                      const int32_t* __restrict__ front_idx,
                                                      const double* __restrict__ arena,
                                                      const double* __restrict__ tg, const double* __restrict__ ts,
                                                      double* __restrict__ y, double* __restrict__ uvec) {
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m, k = F.k;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const double* Fm = arena + F.F_off;
    if (j < k) y[front_idx[F.idx_off + j]] = ts[F.idx_off + j] / Fm[j + (int64_t)j * m];
    else uvec[F.u_off + j - k] = tg[F.idx_off + j];
}

// v[q] = y[piv q] - sum_{r >= k} L[r, q] x[bnd r]: one wave per pivot column.
__global__ __launch_bounds__(256) void mf_bwd_big_init(const FrontDev* __restrict__ fr, int32_t first,
                                                       const int32_t* __restrict__ front_idx,
                                                       const double* __restrict__ arena,
                                                       const double* __restrict__ y, const double* __restrict__ x,
                                                       double* __restrict__ tg) {
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m, k = F.k;
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= k) return;
    const int32_t* idx = front_idx + F.idx_off;
    const double* Lq = arena + F.F_off + (int64_t)q * m;
    double s = 0.0;
    for (int r = k + lane; r < m; r += 64) s += Lq[r] * x[idx[r]];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) tg[F.idx_off + q] = y[idx[q]] - s;
}

// Block step (descending j0): solve the unit-upper diagonal block, publish x, and update the
// entries q < j0 with rows j0..j0+nb of L (32 contiguous doubles per column).
__global__ __launch_bounds__(256) void mf_bwd_big_step(const FrontDev* __restrict__ fr, int32_t first, int j0,
                                                       const int32_t* __restrict__ front_idx,
                                                       const double* __restrict__ arena, double* __restrict__ tg,
                                                       double* __restrict__ x) {
    __shared__ double Dk[NB][NB + 1];
    __shared__ double xb[NB];
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m, k = F.k;
    if (j0 >= k) return;
    const int nb = min(NB, k - j0);
    const int q0 = blockIdx.x * 256;
    if (blockIdx.x > 0 && q0 >= j0) return;
    const double* Fm = arena + F.F_off;
    double* t = tg + F.idx_off;
    const int tid = threadIdx.x;
    for (int i = tid; i < nb * nb; i += 256) {
        const int r = i % nb, c = i / nb;
        Dk[r][c] = (r > c) ? Fm[(j0 + r) + (int64_t)(j0 + c) * m] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        double v = (tid < nb) ? t[j0 + tid] : 0.0;
#pragma unroll
        for (int c = NB - 1; c >= 0; --c) {
            if (c < nb) {
                const double xc = readlane_f64(v, c);
                if (tid < c) v -= Dk[c][tid] * xc;
            }
        }
        if (tid < nb) {
            xb[tid] = v;
            if (blockIdx.x == 0) x[front_idx[F.idx_off + j0 + tid]] = v;
        }
    }
    __syncthreads();
    const int q = q0 + tid;
    if (q < j0) {
        const double* Lq = Fm + (int64_t)q * m + j0;
        double v = t[q];
        for (int c = 0; c < nb; ++c) v -= Lq[c] * xb[c];
        t[q] = v;
    }
}


// ---- leaf fronts with m <= 16 (the static-condensation leaves: one per element) -----------------
// A wave-per-front kernel leaves 3/4 of its lanes idle on these and pays a full LDS instruction
// per handful of entries.  Here 16 lanes own one front (4 fronts per wave, 16 per workgroup):
// lane r keeps row r of the front in registers, the column of multipliers is exchanged through a
// 16-double LDS line per front, and the triangular solves use width-16 shuffles.
__global__ __launch_bounds__(256) void mf_factor_tiny(const FrontDev* __restrict__ fr, int32_t first, int32_t count,
                                                      const int32_t* __restrict__ a_src,
                                                      const int32_t* __restrict__ a_dst,
                                                      const double* __restrict__ Hval, double* __restrict__ arena,
                                                      int32_t* __restrict__ status) {
    __shared__ double Wt[16][136];          // packed lower triangle, column stride 16 (a_dst is remapped by analyze())
    __shared__ double colb[16][16];
    const int g = threadIdx.x >> 4, r = threadIdx.x & 15;
    const int fi = blockIdx.x * 16 + g;
    const bool on = fi < count;
    const FrontDev F = fr[first + (on ? fi : 0)];
    const int m = F.m, k = on ? F.k : 0;
    double* W = Wt[g];
    for (int i = r; i < 136; i += 16) W[i] = 0.0;
    wave_sync();
    if (on)
        for (int t = r; t < F.a_cnt; t += 16) W[a_dst[F.a_off + t]] = Hval[a_src[F.a_off + t]];
    wave_sync();
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = (on && r < m && c <= r) ? W[c * 16 - c * (c - 1) / 2 + (r - c)] : 0.0;
    int kmax = k;
    kmax = max(kmax, __shfl_xor(kmax, 16, 64));
    kmax = max(kmax, __shfl_xor(kmax, 32, 64));
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j >= kmax) break;
        colb[g][r] = a[j];
        wave_sync();
        const bool act = j < k;
        const double d = colb[g][j];
        if (act && (d == 0.0 || !isfinite(d))) bad = true;
        const double lr = a[j] * fast_recip(act ? d : 1.0);
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
            const double v = colb[g][c];          // entry (c, j); rows >= m hold zeros
            if (act) a[c] -= lr * v;
        }
        if (act && r > j) a[j] = lr;
        wave_sync();
    }
    if (bad) atomicOr(status, 1);
    if (on && r < m) {
        double* Fg = arena + F.F_off;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c <= r) Fg[tiny_entry(F, r, c)] = a[c];
    }
}

__global__ __launch_bounds__(256) void mf_forward_tiny(const FrontDev* __restrict__ fr, int32_t first, int32_t count,
                                                       const int32_t* __restrict__ front_idx,
                                                       const double* __restrict__ arena,
                                                       const double* __restrict__ b, double* __restrict__ y,
                                                       double* __restrict__ uvec) {
    const int g = threadIdx.x >> 4, r = threadIdx.x & 15;
    const int fi = blockIdx.x * 16 + g;
    const bool on = fi < count;
    const FrontDev F = fr[first + (on ? fi : 0)];
    const int m = F.m, k = on ? F.k : 0;
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    const bool row = on && r < m;
    const int myidx = row ? idx[r] : 0;
    double t = (row && r < k) ? b[myidx] : 0.0;
    double l[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) l[j] = (row && j < k && r > j) ? Fm[tiny_entry(F, r, j)] : 0.0;
    const double dr = (row && r < k) ? Fm[tiny_entry(F, r, r)] : 1.0;
    int kmax = k;
    kmax = max(kmax, __shfl_xor(kmax, 16, 64));
    kmax = max(kmax, __shfl_xor(kmax, 32, 64));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j >= kmax) break;
        const double tj = __shfl(t, j, 16);
        t -= l[j] * tj;                            // l[j] = 0 outside (j < k, r > j)
    }
    if (row) {
        if (r < k) y[myidx] = t / dr;
        else uvec[F.u_off + r - k] = t;
    }
}

__global__ __launch_bounds__(256) void mf_backward_tiny(const FrontDev* __restrict__ fr, int32_t first, int32_t count,
                                                        const int32_t* __restrict__ front_idx,
                                                        const double* __restrict__ arena,
                                                        const double* __restrict__ y, double* __restrict__ x) {
    const int g = threadIdx.x >> 4, r = threadIdx.x & 15;
    const int fi = blockIdx.x * 16 + g;
    const bool on = fi < count;
    const FrontDev F = fr[first + (on ? fi : 0)];
    const int m = F.m, k = on ? F.k : 0;
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    const bool row = on && r < m;
    const int myidx = row ? idx[r] : 0;
    double t = row ? ((r < k) ? y[myidx] : x[myidx]) : 0.0;
    double l[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) l[j] = (row && j < k && r > j) ? Fm[tiny_entry(F, r, j)] : 0.0;
    int kmax = k;
    kmax = max(kmax, __shfl_xor(kmax, 16, 64));
    kmax = max(kmax, __shfl_xor(kmax, 32, 64));
#pragma unroll
    for (int j = 15; j >= 0; --j) {
        if (j >= kmax) continue;
        double s = l[j] * t;                       // rows r > j of column j (zero elsewhere)
        s = row16_sum_f64(s);
        if (r == j && j < k) t -= s;
    }
    if (row && r < k) x[myidx] = t;
}

// ---- large-front triangular solves, one workgroup per front -----------------------------------
// The block steps of a triangular solve are a chain of dependent latencies (diagonal block ->
// row update -> next diagonal block); the arithmetic is tiny.  One 1024-thread workgroup per
// front keeps the whole work vector in LDS and turns every kernel boundary of the multi-launch
// path into a workgroup barrier; fronts of a level run side by side on their own CUs.  L is
// streamed once (coalesced along rows in the forward sweep).  Used while the vector fits in LDS.
constexpr int BIG1_THREADS = 1024;
constexpr int BIG1_MAX_M = 6000;       // work vector + diagonal block within the 64 KB static LDS budget

__global__ __launch_bounds__(BIG1_THREADS) void mf_fwd_big1(const FrontDev* __restrict__ fr, int32_t first,
                                                            const int32_t* __restrict__ front_idx,
                                                            const int32_t* __restrict__ children,
                                                            const int32_t* __restrict__ rel,
                                                            const double* __restrict__ arena,
                                                            const double* __restrict__ b, double* __restrict__ y,
                                                            double* __restrict__ uvec) {
    extern __shared__ double sh[];
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x, nt = BIG1_THREADS;
    double* tl = sh;                         // [m]
    double* Dk = sh + ((m + 1) & ~1);        // [NB][NB + 1]
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    for (int j = tid; j < m; j += nt) tl[j] = (j < k) ? b[idx[j]] : 0.0;
    __syncthreads();
    for (int c = 0; c < F.nchild; ++c) {
        const FrontDev C = fr[children[F.child_off + c]];
        const int32_t* rl = rel + C.rel_off;
        const double* uc = uvec + C.u_off;
        const int bc = C.m - C.k;
        for (int j = tid; j < bc; j += nt) tl[rl[j]] += uc[j];
        __syncthreads();
    }
    // strictly-lower entry (r, c) of the first diagonal block, one per thread
    const int dr = tid % NB, dc = tid / NB;
    double dnext = (dr > dc && dr < k && dc < k) ? Fm[dr + (int64_t)dc * m] : 0.0;
    for (int j0 = 0; j0 < k; j0 += NB) {
        const int nb = min(NB, k - j0);
        Dk[dr * (NB + 1) + dc] = dnext;
        __syncthreads();
        {   // prefetch the next diagonal block while this one is used
            const int jn = j0 + NB;
            dnext = (dr > dc && jn + dr < k && jn + dc < k) ? Fm[(jn + dr) + (int64_t)(jn + dc) * m] : 0.0;
        }
        if (tid < 64) {
            double v = (tid < nb) ? tl[j0 + tid] : 0.0;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const double tc = readlane_f64(v, c);
                if (tid > c && tid < NB) v -= Dk[tid * (NB + 1) + c] * tc;      // rows/columns >= nb hold zeros
            }
            if (tid < nb) tl[j0 + tid] = v;
        }
        __syncthreads();
        for (int r = j0 + nb + tid; r < m; r += nt) {
            const double* Lr = Fm + r + (int64_t)j0 * m;
            double v = tl[r];
            if (nb == NB) {
#pragma unroll
                for (int c = 0; c < NB; ++c) v -= Lr[(int64_t)c * m] * tl[j0 + c];
            } else {
                for (int c = 0; c < nb; ++c) v -= Lr[(int64_t)c * m] * tl[j0 + c];
            }
            tl[r] = v;
        }
        __syncthreads();
    }
    for (int j = tid; j < m; j += nt) {
        if (j < k) y[idx[j]] = tl[j] / Fm[j + (int64_t)j * m];
        else uvec[F.u_off + j - k] = tl[j];
    }
}

__global__ __launch_bounds__(BIG1_THREADS) void mf_bwd_big1(const FrontDev* __restrict__ fr, int32_t first,
                                                            const int32_t* __restrict__ front_idx,
                                                            const double* __restrict__ arena,
                                                            const double* __restrict__ y, double* __restrict__ x) {
    extern __shared__ double sh[];
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x, nt = BIG1_THREADS;
    const int lane = tid & 63, wave = tid >> 6;
    double* tl = sh;                         // [m]: pivots hold the running right-hand side, the rest x(boundary)
    double* Dk = sh + ((m + 1) & ~1);        // [NB][NB + 1]
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    for (int j = tid; j < m; j += nt) tl[j] = (j < k) ? y[idx[j]] : x[idx[j]];
    __syncthreads();
    // v[q] = y[q] - sum_{r >= k} L[r, q] x[r]: one wave per pivot column
    for (int q = wave; q < k; q += nt / 64) {
        const double* Lq = Fm + (int64_t)q * m;
        double s = 0.0;
        for (int r = k + lane; r < m; r += 64) s += Lq[r] * tl[r];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) tl[q] -= s;
    }
    __syncthreads();
    const int dr = tid % NB, dc = tid / NB;
    const int last = ((k - 1) / NB) * NB;
    double dnext = (dr > dc && last + dr < k) ? Fm[(last + dr) + (int64_t)(last + dc) * m] : 0.0;
    for (int j0 = last; j0 >= 0; j0 -= NB) {
        const int nb = min(NB, k - j0);
        Dk[dr * (NB + 1) + dc] = dnext;
        __syncthreads();
        if (j0 >= NB) {
            const int jn = j0 - NB;          // full block
            dnext = (dr > dc) ? Fm[(jn + dr) + (int64_t)(jn + dc) * m] : 0.0;
        }
        if (tid < 64) {
            double v = (tid < nb) ? tl[j0 + tid] : 0.0;
#pragma unroll
            for (int c = NB - 1; c >= 0; --c) {
                const double xc = readlane_f64(v, c);
                if (tid < c) v -= Dk[c * (NB + 1) + tid] * xc;      // rows/columns >= nb hold zeros
            }
            if (tid < nb) {
                tl[j0 + tid] = v;
                x[idx[j0 + tid]] = v;
            }
        }
        __syncthreads();
        for (int q = tid; q < j0; q += nt) {
            const double* Lq = Fm + (int64_t)q * m + j0;
            double v = tl[q];
            if (nb == NB) {
#pragma unroll
                for (int c = 0; c < NB; ++c) v -= Lq[c] * tl[j0 + c];
            } else {
                for (int c = 0; c < nb; ++c) v -= Lq[c] * tl[j0 + c];
            }
            tl[q] = v;
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------
// large fronts, inverse-based path (fronts with 128 < m <= BIG_INV_MAX_M)
// ------------------------------------------------------------------------------------------------
// One launch per 32-column step.  The diagonal block of step j is factored AND inverted ahead of
// time by the look-ahead workgroup of step j-1 (W_j = L_jj^{-1}, d_j); every trailing tile then
// forms the two panel slices it needs by a small matrix-core product with W_j,
//     S = A21 W_j'  (= L21 D),   L = S D^{-1},
// instead of waiting for a separate triangular-solve kernel, and applies  C -= S L'  on the matrix
// cores (v_mfma_f64_16x16x4_f64).  The panel itself is never written back: the arena keeps the
// fully updated, UNSOLVED rows A21, and the triangular sweeps need one matrix per block,
// M_j = W_j' D_j^{-1} W_j (the inverse of the updated diagonal block):  forward u_j = M_j t_j,
// t_r -= A_rj u_j;  backward x_j = u_j - M_j G_j with G = A21' x_r.  Home layout of a factored
// diagonal block: strictly UPPER triangle = off-diagonal of M_j, its diagonal lives in `dvec`;
// the lower triangle keeps the unfactored block (sibling workgroups of step 0 still read it).
constexpr int BIG_INV_MAX_M = 7000;     // work vectors of the single-workgroup solves stay in LDS
constexpr int BIGI_THREADS = 1024;


#include "ldlt32.hpp"

// The 32 x 32 LDL' of the pivot chain and the inverse of its factor: one wave each, every product on the matrix cores
// (ldlt32.hpp; the 256-thread forms of round 2 -- 4 x 4-blocked LDL', seven-barrier recursive doubling -- were removed in
// round 4: no build selected them).
__device__ __forceinline__ void block_ldlt32(double (*Dn)[NB + 1], double* dq, int nb, int tid, int32_t* __restrict__ status) {
    block_ldlt32_mfma(Dn, dq, nb, tid, status);
}
__device__ __forceinline__ void block_inverse32_sel(const double (*Ls)[NB + 1], double (*Wv)[NB + 1], double (*Tm)[17], int tid) {
    block_inverse32_mfma(Ls, Wv, Tm, tid);
}

// S = A W' for a 64-row slice held raw in P[c][rr] (LDS, overwritten in place); wave w owns rows
// 16w .. 16w+15, so no cross-wave hazard.  scale != nullptr: result columns are multiplied by
// scale[q] (the reciprocal pivots) and written to Pout (may alias P).
__device__ __forceinline__ void slice_transform(double (*P)[ST + 1], double (*Pout)[ST + 1], double (*P2)[ST + 1],
                                                const double (*Wv)[NB + 1], const double* rd, int lane, int wave) {
    const int fr = lane & 15, fk = lane >> 4;
    const int rr = 16 * wave + fr;
    double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
        const double a = P[4 * kk + fk][rr];                       // y[k][j]: A[rr = j][c = k]
        if (kk < 4) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Wv[fr][4 * kk + fk], a, acc0, 0, 0, 0);   // q tile 0: c < 16 only
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Wv[16 + fr][4 * kk + fk], a, acc1, 0, 0, 0);
    }
    // D[i][j]: i = fk + 4 reg -> q within the tile, j = fr -> rr
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q0 = fk + 4 * r, q1 = 16 + fk + 4 * r;
        Pout[q0][rr] = acc0[r];
        Pout[q1][rr] = acc1[r];
        if (P2) {
            P2[q0][rr] = acc0[r] * rd[q0];
            P2[q1][rr] = acc1[r] * rd[q1];
        }
    }
}

#ifdef MGB_STEP_PROBE      // development probe build only (tools/gpu_probe.py): per-phase timestamps of one step
#define PROBE(i) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); if (is_la && blockIdx.y == 0 && tid == 0 && j0 == 64 && gridDim.y == 1 && F.k > 400) g_probe[i] = wall_clock64(); if (!is_la && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && j0 == 64 && gridDim.y == 1 && F.k > 400) g_probe[16 + i] = wall_clock64(); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PROBE(i) do { } while (0)
#endif

__global__ __launch_bounds__(256, 3) void mf_big_step(const FrontDev* __restrict__ fr, int32_t first, int j0,
                                                   double* __restrict__ arena, double* __restrict__ dscr,
                                                   double* __restrict__ dvec, int32_t* __restrict__ status,
                                                   int do_diag) {
    __shared__ double Wv[NB][NB + 1];
    __shared__ double Dn[NB][NB + 1];
    __shared__ double Tm[16][17];
    __shared__ double dq[NB], rdq[NB];
    __shared__ double Pa[NB][ST + 1];
    __shared__ double Pb[NB][ST + 1];
    const FrontDev F = fr[first + blockIdx.y];      // (as kernel arguments for launches of few fronts: the kernel sits at its
                                                      // 168-register cap and spilled, 6 % slower end to end: measured in round 4)
    const int m = F.m, k = F.k;
    if (j0 >= k) return;
    const int nb = min(NB, k - j0);
    const int j1 = j0 + nb;
    const int T = (m - j1 + ST - 1) / ST;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* W = arena + F.F_off;
    double* slot = dscr + ((int64_t)blockIdx.y * 2 + ((j0 / NB) & 1)) * (NB * NB);
    const bool is_la = blockIdx.x == gridDim.x - 1;
    int ti = 0, tj = 0;
    if (!is_la) {
        const int lin = blockIdx.x;
        ti = (int)((sqrt(8.0 * lin + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= lin) ++ti;
        while (ti * (ti + 1) / 2 > lin) --ti;
        tj = lin - ti * (ti + 1) / 2;
        if (ti >= T) return;
    }
    const bool look = j1 < k;
    const int nbn = look ? min(NB, k - j1) : 0;
    const int rbase = is_la ? j1 : j1 + ti * ST, cbase = j1 + tj * ST;
    PROBE(0);
#ifdef MGB_STEP_PROBE
    if (is_la && blockIdx.y == 0 && tid == 0 && j0 == 64 && gridDim.y == 1 && F.k > 400) g_probe[34] = clock64();
#endif
    const int fr16 = lane & 15, fk = lane >> 4;

    // ---- global loads first: raw panel slices and this wave's part of the C tile -----------------
    double pa[NB * ST / 256], pb[NB * ST / 256];
#pragma unroll
    for (int u = 0; u < NB * ST / 256; ++u) {
        const int i = tid + 256 * u, rr = i % ST, q = i / ST;
        const int r = rbase + rr, c = cbase + rr;
        const bool rin = is_la ? (rr < nbn) : (r < m);
        pa[u] = (q < nb && rin) ? W[r + (int64_t)(j0 + q) * m] : 0.0;
        pb[u] = (!is_la && ti != tj && q < nb && c < m) ? W[c + (int64_t)(j0 + q) * m] : 0.0;
    }
    double cw[4][4];
    if (!is_la) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rbase + 16 * wave + fr16, col = cbase + 16 * tb + fk + 4 * r;
                cw[tb][r] = (row < m && col < m && row >= col) ? W[row + (int64_t)col * m] : 0.0;
            }
    } else {
        // corner of the next diagonal block, entry (rr, c) per thread x 4
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = tid + 256 * t, rr = i % NB, c = i / NB;
            cw[0][t] = (look && rr >= c && rr < nbn) ? W[(j1 + rr) + (int64_t)(j1 + c) * m] : 0.0;
        }
    }
    // ---- W_j, d_j ----------------------------------------------------------------------------------
    if (do_diag) {
        for (int i = tid; i < NB * NB; i += 256) {
            const int rr = i % NB, c = i / NB;
            Dn[rr][c] = (rr >= c && rr < nb) ? W[(j0 + rr) + (int64_t)(j0 + c) * m] : 0.0;
        }
        __syncthreads();
        block_ldlt32(Dn, dq, nb, tid, is_la ? status : nullptr);
        block_inverse32_sel(Dn, Wv, Tm, tid);
    } else {
        for (int i = tid; i < NB * NB; i += 256) {
            const int rr = i % NB, c = i / NB;
            const double v = (rr >= c && rr < nb) ? slot[rr + NB * c] : 0.0;
            Wv[rr][c] = (rr > c) ? v : (rr == c ? 1.0 : 0.0);
            if (rr == c) dq[rr] = (rr < nb) ? v : 1.0;
        }
    }
#pragma unroll
    for (int u = 0; u < NB * ST / 256; ++u) {
        const int i = tid + 256 * u, rr = i % ST, q = i / ST;
        Pa[q][rr] = pa[u];
        Pb[q][rr] = pb[u];
    }
    __syncthreads();
    PROBE(1);
    if (tid < NB) rdq[tid] = 1.0 / dq[tid];
    // home of block j (nobody reads it during this step): M_j = W_j' D_j^{-1} W_j, the inverse of the updated
    // diagonal block -- the triangular sweeps need nothing else of the block (mf_fwd_inv / mf_bwd_inv).
    // Strictly upper triangle = off-diagonal of M_j, diagonal of M_j to dvec.  One 16 x 16 tile per wave on the
    // matrix cores, straight from the accumulators (the tile above the diagonal is the mirror image: skipped).
    // Written by tile workgroup 0 (done at 6 us, every workgroup has W_j staged) rather than by the look-ahead
    // workgroup, whose 1.7 us for it sat on the critical path of the pivot chain; steps without tiles keep it there.
    if (is_la ? T == 0 : blockIdx.x == 0) {          // T is this front's own tile count (a batch is launched for its largest front)
        {
            const int rt = wave & 1, ct = wave >> 1;
            if (ct <= rt) {
                double4_t accm = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < NB / 4; ++kk) {
                    const int kq = 4 * kk + fk;
                    accm = __builtin_amdgcn_mfma_f64_16x16x4f64(kq < nb ? Wv[kq][16 * ct + fr16] / dq[kq] : 0.0, Wv[kq][16 * rt + fr16], accm, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * rt + fr16, col = 16 * ct + fk + 4 * r;
                    if (row < nb && col < row) W[(j0 + col) + (int64_t)(j0 + row) * m] = accm[r];
                    else if (row < nb && col == row) dvec[F.idx_off + j0 + row] = accm[r];
                }
            }
        }
    }
    if (is_la && !look) return;
    __syncthreads();
    PROBE(2);
    // ---- S (into Pa) and L (into Pb) ---------------------------------------------------------------
    if (is_la) {
        if (wave < 2) slice_transform(Pa, Pa, Pb, Wv, rdq, lane, wave);     // 32 rows of the next block
        __syncthreads();
        PROBE(3);
        {   // D_{j+1} = corner - S L' on the matrix cores: wave w -> (row tile w & 1, column tile w >> 1)
            const int rt = wave & 1, ct = wave >> 1;
            double4_t accd = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < NB / 4; ++kk)
                accd = __builtin_amdgcn_mfma_f64_16x16x4f64(Pb[4 * kk + fk][16 * ct + fr16], Pa[4 * kk + fk][16 * rt + fr16],
                                                            accd, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * rt + fr16, c = 16 * ct + fk + 4 * r;
                Dn[rr][c] = -accd[r];        // the corner entries are added by their loader threads below
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = tid + 256 * t, rr = i % NB, c = i / NB;
            Dn[rr][c] = (rr >= c && rr < nbn) ? cw[0][t] + Dn[rr][c] : 0.0;
        }
        __syncthreads();
        double* nslot = dscr + ((int64_t)blockIdx.y * 2 + ((j1 / NB) & 1)) * (NB * NB);
        PROBE(4);
#ifdef MGB_STEP_PROBE
        // cold / warm experiment: the same factorization twice (Dn saved and restored in between)
        double sv[4];
        for (int t = 0; t < 4; ++t) { const int i = tid + 256 * t; sv[t] = Dn[i % NB][i / NB]; }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (is_la && blockIdx.y == 0 && tid == 0 && j0 == 64 && gridDim.y == 1 && F.k > 400) { g_probe[32] = clock64(); g_probe[36] = wall_clock64(); }
        __builtin_amdgcn_sched_barrier(0);
        block_ldlt32(Dn, dq, nbn, tid, nullptr);
        __builtin_amdgcn_sched_barrier(0);
        if (is_la && blockIdx.y == 0 && tid == 0 && j0 == 64 && gridDim.y == 1 && F.k > 400) { g_probe[33] = clock64(); g_probe[37] = wall_clock64(); }
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < 4; ++t) { const int i = tid + 256 * t; Dn[i % NB][i / NB] = sv[t]; }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (is_la && blockIdx.y == 0 && tid == 0 && j0 == 64 && gridDim.y == 1 && F.k > 400) { g_probe[38] = clock64(); g_probe[44] = wall_clock64(); }
        __builtin_amdgcn_sched_barrier(0);
#endif
        block_ldlt32(Dn, dq, nbn, tid, status);
#ifdef MGB_STEP_PROBE
        __builtin_amdgcn_sched_barrier(0);
        if (is_la && blockIdx.y == 0 && tid == 0 && j0 == 64 && gridDim.y == 1 && F.k > 400) { g_probe[39] = clock64(); g_probe[45] = wall_clock64(); }
        __builtin_amdgcn_sched_barrier(0);
#endif
        PROBE(5);
        block_inverse32_sel(Dn, Wv, Tm, tid);
        PROBE(6);
        for (int i = tid; i < NB * NB; i += 256) {        // slot: diagonal d, strictly lower W (column-major)
            const int rr = i % NB, c = i / NB;
            if (rr >= c && rr < nbn) nslot[rr + NB * c] = (rr == c) ? dq[rr] : Wv[rr][c];
        }
        PROBE(7);
#ifdef MGB_STEP_PROBE
        if (is_la && blockIdx.y == 0 && tid == 0 && j0 == 64 && gridDim.y == 1 && F.k > 400) g_probe[35] = clock64();
#endif
        return;
    }
    if (ti == tj) {
        slice_transform(Pa, Pa, Pb, Wv, rdq, lane, wave);
    } else {
        slice_transform(Pa, Pa, nullptr, Wv, rdq, lane, wave);
        // the column-side slice: L = (A W') D^{-1}
        {
            const int rr = 16 * wave + fr16;
            double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < NB / 4; ++kk) {
                const double a = Pb[4 * kk + fk][rr];
                if (kk < 4) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Wv[fr16][4 * kk + fk], a, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Wv[16 + fr16][4 * kk + fk], a, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q0 = fk + 4 * r, q1 = 16 + fk + 4 * r;
                Pb[q0][rr] = acc0[r] * rdq[q0];
                Pb[q1][rr] = acc1[r] * rdq[q1];
            }
        }
    }
    __syncthreads();
    PROBE(3);
    // ---- C -= S L' : wave w owns rows 16w..16w+15, four 16-column tiles ---------------------------
    double4_t acc[4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) acc[tb] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
        const double s = Pa[4 * kk + fk][16 * wave + fr16];            // y[k][j]: S[rr = j][q = k]
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            if (ti == tj && tb > wave) continue;                          // strictly above the diagonal
            acc[tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pb[4 * kk + fk][16 * tb + fr16], s, acc[tb], 0, 0, 0);
        }
    }
    const int nskip = (blockIdx.x == 0 && look) ? nbn : 0;               // corner owned by the look-ahead workgroup
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rbase + 16 * wave + fr16, col = cbase + 16 * tb + fk + 4 * r;
            if (row < m && col < m && row >= col && !(row - j1 < nskip && col - j1 < nskip))
                W[row + (int64_t)col * m] = cw[tb][r] - acc[tb][r];
        }
    PROBE(4);
}

// Interface front of a domain-decomposed system: only its lower triangle is meaningful, so only that crosses ranks --
// packed column by column (column c at c*m - c(c-1)/2, rows c .. m-1), summed, unpacked in place.
__global__ __launch_bounds__(256) void mf_tri_pack(int m, const double* __restrict__ F, double* __restrict__ packed, int unpack,
                                                   double* __restrict__ Fout) {
    const int c = blockIdx.x;
    if (c >= m) return;
    const int64_t base = (int64_t)c * m - ((int64_t)c * (c - 1)) / 2;
    for (int r = c + threadIdx.x; r < m; r += 256) {
        if (unpack) Fout[r + (int64_t)c * m] = packed[base + (r - c)];
        else packed[base + (r - c)] = F[r + (int64_t)c * m];
    }
}

// First diagonal block of every front of a batch, factored and inverted once (one workgroup per front)
// into slot 0.  Used for batches of many fronts, where the redundant factorization inside every trailing
// tile of step 0 (do_diag) would occupy all compute units with copies of the same 32 x 32 problem.
__global__ __launch_bounds__(256) void mf_big_diag0(const FrontDev* __restrict__ fr, int32_t first,
                                                    const double* __restrict__ arena, double* __restrict__ dscr,
                                                    int32_t* __restrict__ status) {
    __shared__ double Wv[NB][NB + 1];
    __shared__ double Dn[NB][NB + 1];
    __shared__ double Tm[16][17];
    __shared__ double dq[NB];
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, nb = min(NB, F.k), tid = threadIdx.x;
    const double* W = arena + F.F_off;
    for (int i = tid; i < NB * NB; i += 256) {
        const int rr = i % NB, c = i / NB;
        Dn[rr][c] = (rr >= c && rr < nb) ? W[rr + (int64_t)c * m] : 0.0;
    }
    __syncthreads();
    block_ldlt32(Dn, dq, nb, tid, status);
    block_inverse32_sel(Dn, Wv, Tm, tid);
    double* slot = dscr + (int64_t)blockIdx.x * 2 * (NB * NB);
    for (int i = tid; i < NB * NB; i += 256) {
        const int rr = i % NB, c = i / NB;
        if (rr >= c && rr < nb) slot[rr + NB * c] = (rr == c) ? dq[rr] : Wv[rr][c];
    }
}

// Gather form of the assembly for fronts with few children (every front of a nested-dissection tree above the
// leaves has two to four): the inverse of every child's relative index list is laid out in LDS, then each
// destination entry is formed ONCE in a register -- the children's entries that land on it, added in child order,
// all their loads in flight together -- and stored once.  No zero pass, no read-modify-write of the arena, and
// the dependent-load chains of the children run side by side instead of one child after the other.
constexpr int GATHER_MAX_CHILD = 8;
__global__ __launch_bounds__(256) void mf_big_gather(const FrontDev* __restrict__ fr, int32_t first,
                                                     const int32_t* __restrict__ children,
                                                     const int32_t* __restrict__ rel,
                                                     const int32_t* __restrict__ a_src,
                                                     const int32_t* __restrict__ a_dst,
                                                     const int32_t* __restrict__ a_colptr,
                                                     const double* __restrict__ Hval, double* __restrict__ arena, int mstride,
                                                     double* __restrict__ dscr, int32_t* __restrict__ status, int with_diag,
                                                     int ct /* destination columns per workgroup */) {
    extern __shared__ int32_t inv[];               // [nchild][mstride]: position in the child's update block or -1
    __shared__ int64_t cU[GATHER_MAX_CHILD];
    __shared__ int64_t cR[GATHER_MAX_CHILD];
    __shared__ int32_t cM[GATHER_MAX_CHILD], cB[GATHER_MAX_CHILD];
    const FrontDev F = fr[first + blockIdx.y];
    const int m = F.m;
    if (with_diag && blockIdx.x == gridDim.x - 1) {
        // One extra workgroup per front forms ONLY the first 32 x 32 diagonal block (same gather, same order as
        // the column workgroups, which write it to the arena), factors and inverts it and leaves W_0 / d_0 in slot 0:
        // step 0 of the factorization finds its diagonal block ready, as every later step does from the look-ahead
        // workgroup.  Its time hides under the column workgroups of the same launch (was: a launch of its own for
        // batches of many fronts, a redundant factorization inside every tile of step 0 for the others).
        __shared__ double Wv[NB][NB + 1];
        __shared__ double Dn[NB][NB + 1];
        __shared__ double Tm[16][17];
        __shared__ double dq[NB];
            __shared__ int32_t inv0[GATHER_MAX_CHILD][NB];
        const int tid = threadIdx.x, nch = F.nchild, nb = min(NB, F.k);
        // the first batch of A entries of the block (cp -> a_dst / a_src -> Hval: three dependent loads) is requested before
        // the children's chain (children -> descriptor -> rel -> arena: four more) instead of after it
        int a_d0 = -1;
        double a_v0 = 0.0;
        const int a_end = (a_colptr + F.acol_off)[nb];
        if (tid < a_end) {
            a_d0 = a_dst[F.a_off + tid];
            a_v0 = Hval[a_src[F.a_off + tid]];
        }
        if (tid < nch) {
            const FrontDev C = fr[children[F.child_off + tid]];
            cU[tid] = C.F_off + (int64_t)C.k * C.m + C.k;
            cR[tid] = C.rel_off;
            cM[tid] = C.m;
            cB[tid] = C.m - C.k;
        }
        for (int i = tid; i < GATHER_MAX_CHILD * NB; i += 256) inv0[i / NB][i % NB] = -1;
        __syncthreads();
        for (int ch = 0; ch < nch; ++ch) {               // rel is increasing: only its first entries can be < 32
            const int32_t* rl = rel + cR[ch];
            const int lim = min(cB[ch], NB);
            if (tid < lim) {
                const int g = rl[tid];
                if (g < NB) inv0[ch][g] = tid;
            }
        }
        __syncthreads();
        for (int i = tid; i < NB * NB; i += 256) {
            const int rr = i % NB, c = i / NB;
            double v = 0.0;
            if (rr >= c && rr < nb) {
                for (int ch = 0; ch < nch; ++ch) {        // child order: the summation order of the extend-add
                    const int jc = inv0[ch][c], ir = inv0[ch][rr];
                    if (jc >= 0 && ir >= 0) v += arena[cU[ch] + (int64_t)jc * cM[ch] + ir];
                }
            }
            Dn[rr][c] = v;
        }
        __syncthreads();
        {
            const int32_t* ad = a_dst + F.a_off;
            if (a_d0 >= 0) {
                const int lu = a_d0 % m, lv = a_d0 / m;
                if (lu < nb) Dn[lu][lv] += a_v0;
            }
            for (int t = tid + 256; t < a_end; t += 256) {
                const int d = ad[t], lu = d % m, lv = d / m;
                if (lu < nb) Dn[lu][lv] += Hval[a_src[F.a_off + t]];
            }
        }
        __syncthreads();
        block_ldlt32(Dn, dq, nb, tid, status);
        block_inverse32_sel(Dn, Wv, Tm, tid);
        double* slot = dscr + (int64_t)blockIdx.y * 2 * (NB * NB);
        for (int i = tid; i < NB * NB; i += 256) {
            const int rr = i % NB, c = i / NB;
            if (rr >= c && rr < nb) slot[rr + NB * c] = (rr == c) ? dq[rr] : Wv[rr][c];
        }
        return;
    }
    const int c0 = blockIdx.x * ct;
    if (c0 >= m) return;
    const int c1 = min(c0 + ct, m);
    double* W = arena + F.F_off;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;     // one wave per destination column, lanes on the rows
    const int nch = F.nchild;
    if (tid < nch) {
        const FrontDev C = fr[children[F.child_off + tid]];
        cU[tid] = C.F_off + (int64_t)C.k * C.m + C.k;
        cR[tid] = C.rel_off;
        cM[tid] = C.m;
        cB[tid] = C.m - C.k;
    }
    for (int i = tid; i < nch * mstride; i += 256) inv[i] = -1;
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int32_t* rl = rel + cR[ch];
        const int b = cB[ch];
        for (int j = tid; j < b; j += 256) inv[ch * mstride + rl[j]] = j;
    }
    __syncthreads();
    for (int c = c0 + wave; c < c1; c += 4) {
        double* Wc = W + (int64_t)c * m;
        int64_t colbase[GATHER_MAX_CHILD];         // child column offset, -1 when the child does not reach column c
#pragma unroll
        for (int ch = 0; ch < GATHER_MAX_CHILD; ++ch) {
            const int jc = ch < nch ? inv[ch * mstride + c] : -1;
            colbase[ch] = jc >= 0 ? cU[ch] + (int64_t)jc * cM[ch] : -1;
        }
        for (int r = c + lane; r < m; r += 128) {  // two rows per lane in flight (four: 2 % slower end to end, measured in round 4)
            const int r1 = r + 64;
            double u0[GATHER_MAX_CHILD], u1[GATHER_MAX_CHILD];
#pragma unroll
            for (int ch = 0; ch < GATHER_MAX_CHILD; ++ch) {
                u0[ch] = 0.0;
                u1[ch] = 0.0;
                if (colbase[ch] >= 0) {
                    const int i0 = inv[ch * mstride + r];
                    const int i1 = r1 < m ? inv[ch * mstride + r1] : -1;
                    if (i0 >= 0) u0[ch] = arena[colbase[ch] + i0];
                    if (i1 >= 0) u1[ch] = arena[colbase[ch] + i1];
                }
            }
            double v0 = 0.0, v1 = 0.0;
#pragma unroll
            for (int ch = 0; ch < GATHER_MAX_CHILD; ++ch) {      // child order: the summation order of the extend-add
                v0 += u0[ch];
                v1 += u1[ch];
            }
            Wc[r] = v0;
            if (r1 < m) Wc[r1] = v1;
        }
    }
    __syncthreads();
    {   // A entries are grouped by pivot column: the per-column offsets give the range of [c0, c1)
        const int32_t* cp = a_colptr + F.acol_off;
        const int beg = cp[min(c0, F.k)], end = cp[min(c1, F.k)];
        const int32_t* ad = a_dst + F.a_off;
        for (int t = beg + tid; t < end; t += 256) W[ad[t]] += Hval[a_src[F.a_off + t]];
    }
}

// ---- triangular solves on the inverse-based layout: one workgroup per front --------------------
// forward, block j of a front:  u_j = M_j t_j,  t[r] -= A[r, j] u_j  (r below);  the stored intermediate is u
// (M_j = A_jj^{-1} of the updated diagonal block = W_j' D_j^{-1} W_j, written home by mf_big_step)
__global__ __launch_bounds__(BIGI_THREADS) void mf_fwd_inv(const FrontDev* __restrict__ fr, int32_t first,
                                                           const int32_t* __restrict__ front_idx,
                                                           const int64_t* __restrict__ ug_ptr,
                                                           const int64_t* __restrict__ ug_src,
                                                           const double* __restrict__ arena,
                                                           const double* __restrict__ dvec,
                                                           const double* __restrict__ b, double* __restrict__ y,
                                                           double* __restrict__ uvec) {
    extern __shared__ double sh[];
#ifdef MGB_STEP_PROBE
#define FP(i) do { if (threadIdx.x == 0 && gridDim.x > 150) { const long long _t = wall_clock64(); if (blockIdx.x == 0) g_probe[48 + i] = _t; if (i == 0) atomicMin((unsigned long long*)&g_probe[56], (unsigned long long)_t); if (i == 5) { atomicMax((unsigned long long*)&g_probe[57], (unsigned long long)_t); atomicAdd((unsigned long long*)&g_probe[58], (unsigned long long)(_t - g_probe[56])); } } } while (0)
#else
#define FP(i) do { } while (0)
#endif
    FP(0);
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x, nt = BIGI_THREADS;
    double* tl = sh;                               // [m]
    double* Ml = sh + ((m + 1) & ~1);              // [NB][NB + 1]: M_j, full symmetric
    double* uq = Ml + NB * (NB + 1);               // [NB]
    double* part = uq + 2 * NB;                    // [BIGI_THREADS] partial sums of the column-split row update
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    const double* dv = dvec + F.idx_off;
    {   // t = [b(piv); 0] + the children's update vectors: one gather per entry, contributions in child order
        const int64_t* up = ug_ptr + F.ug_off;
        for (int j = tid; j < m; j += nt) {
            double v = (j < k) ? b[idx[j]] : 0.0;
            const int64_t e1 = up[j + 1];
            for (int64_t e = up[j]; e < e1; e += 4) {           // four contributions in flight, added in list order
                double a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = (e + u < e1) ? uvec[ug_src[e + u]] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) v += a[u];
            }
            tl[j] = v;
        }
    }
    const int wa = tid % NB, wb = tid / NB;          // M_j[wb][wa] = M_j[wa][wb] sits at (j0 + wa, j0 + wb), wa < wb
    {
        const double w0 = (wa < wb && wb < k) ? Fm[wa + (int64_t)wb * m] : 0.0;
        if (wa < wb) { Ml[wb * (NB + 1) + wa] = w0; Ml[wa * (NB + 1) + wb] = w0; }
        if (tid < NB) Ml[tid * (NB + 1) + tid] = (tid < k) ? dv[tid] : 0.0;
    }
    __syncthreads();
    FP(1);
    for (int j0 = 0; j0 < k; j0 += NB) {
        const int nb = min(NB, k - j0), j1 = j0 + nb;
        // next block's M and the first 16 panel entries of this thread's row update: neither depends on this
        // block's product, so both are requested now and their latency runs under it
        const int jn = j0 + NB;
        const double wnext = (wa < wb && jn + wb < k) ? Fm[(jn + wa) + (int64_t)(jn + wb) * m] : 0.0;
        const double dnext = (tid < NB && jn + tid < k) ? dv[jn + tid] : 0.0;
        const int rows = m - j1;
        int G = 1;
        while (G < 8 && 2 * G * rows <= nt) G *= 2;
        const int cgp = (G > 1 && rows > 0) ? tid / rows : 0, rr = (G > 1 && rows > 0) ? tid - cgp * rows : tid;
        const bool mine = rows > 0 && (G == 1 ? tid < rows : cgp < G);
        const int cstep = G == 1 ? 1 : G;                 // G == 1: columns 0..15 now, 16..31 later; G > 1: all 32 / G columns
        double pa[16], pc[16];
        {
            const double* Ar = Fm + (j1 + rr) + (int64_t)j0 * m;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int col = cgp + u * cstep;
                pa[u] = (mine && col < NB) ? Ar[(int64_t)min(col, nb - 1) * m] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) pc[u] = (mine && G == 1) ? Ar[(int64_t)min(16 + u, nb - 1) * m] : 0.0;
        }
        {   // u = M_j t_j on all 1024 threads: thread (q, c) forms one term, a 32-lane butterfly sums the row
            const int q = tid >> 5, c = tid & 31;
            double pu = Ml[q * (NB + 1) + c] * (c < nb ? tl[j0 + c] : 0.0);
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) pu += __shfl_xor(pu, off, 32);
            if (c == 0) uq[q] = (q < nb) ? pu : 0.0;
        }
        __syncthreads();
        FP(3);
        // M_j is consumed: stage M_{j+1} (the barrier at the end of the step publishes it)
        if (wa < wb) { Ml[wb * (NB + 1) + wa] = wnext; Ml[wa * (NB + 1) + wb] = wnext; }
        if (tid < NB) {
            Ml[tid * (NB + 1) + tid] = dnext;
            if (tid < nb) tl[j0 + tid] = uq[tid];          // the intermediate the backward sweep starts from
        }
        {   // rows below the block: t[r] -= A[r, j0 .. j1) u.  The panel is column-major, so a thread's 32 terms are
            // 32 strided loads; they are issued in groups (a rolled loop waits one memory latency per term, a 32-way
            // unroll spills at 1024 threads), and fronts with few rows split the columns over G thread groups so that
            // all 1024 threads carry loads; the partial sums meet in LDS in a fixed order.
            if (rows > 0 && G == 1) {
                if (mine) {
                    double v = 0.0;
#pragma unroll
                    for (int u = 0; u < 16; ++u) v += pa[u] * uq[u];                  // uq is zero beyond nb
#pragma unroll
                    for (int u = 0; u < 16; ++u) v += pc[u] * uq[16 + u];
                    tl[j1 + tid] -= v;
                }
                for (int r = j1 + tid + nt; r < m; r += nt) {
                    const double* Ar = Fm + r + (int64_t)j0 * m;
                    double v = 0.0;
                    for (int c0 = 0; c0 < nb; c0 += 8) {
                        double a[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) a[u] = Ar[(int64_t)min(c0 + u, nb - 1) * m];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v += a[u] * uq[c0 + u];
                    }
                    tl[r] -= v;
                }
            } else if (rows > 0) {
                if (cgp < G) {
                    double v = 0.0;
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int col = cgp + u * G;
                        v += pa[u] * (col < nb ? uq[col] : 0.0);
                    }
                    part[cgp * rows + rr] = v;
                }
                __syncthreads();
                if (tid < rows) {
                    double v = 0.0;
                    for (int gg = 0; gg < G; ++gg) v += part[gg * rows + tid];
                    tl[j1 + tid] -= v;
                }
            }
        }
        __syncthreads();
        FP(4);
    }
    for (int j = tid; j < m; j += nt) {
        if (j < k) y[idx[j]] = tl[j];
        else uvec[F.u_off + j - k] = tl[j];
    }
    FP(5);
}

// backward:  x_j = u_j - M_j G_j,  G[q] = sum over solved rows r of A[r, q] x[r]
#ifdef MGB_PROBE_BWD       // root front of a sweep: phase timestamps (tools/gpu_probe_bwd.py; build with -DMGB_STEP_PROBE -DMGB_PROBE_BWD:
                           // the slots are shared with the mf_big_step probes)
#define BP(i) do { if (threadIdx.x == 0 && gridDim.x == 1 && fr[first].k > 400) g_probe[(i)] = wall_clock64(); } while (0)
#else
#define BP(i) do { } while (0)
#endif
__global__ __launch_bounds__(BIGI_THREADS) void mf_bwd_inv(const FrontDev* __restrict__ fr, int32_t first,
                                                           const int32_t* __restrict__ front_idx,
                                                           const double* __restrict__ arena,
                                                           const double* __restrict__ dvec,
                                                           const double* __restrict__ y, double* __restrict__ x) {
    extern __shared__ double sh[];
    BP(0);
    const FrontDev F = fr[first + blockIdx.x];
    const int m = F.m, k = F.k;
    const int tid = threadIdx.x, nt = BIGI_THREADS;
    const int lane = tid & 63, wave = tid >> 6;
    double* tl = sh;                               // [m]: u on the pivots (then x), x(boundary) below
    double* gl = sh + ((m + 1) & ~1);              // [k]
    double* Ml = gl + ((k + 1) & ~1);              // [NB][NB + 1]
    double* zq = Ml + NB * (NB + 1);               // [NB]
    int32_t* il = reinterpret_cast<int32_t*>(zq + 2 * NB);    // [m]: the front's index list (x_j is scattered through it)
    const int32_t* idx = front_idx + F.idx_off;
    const double* Fm = arena + F.F_off;
    const double* dv = dvec + F.idx_off;
    for (int j = tid; j < m; j += nt) {
        const int32_t ij = idx[j];
        il[j] = ij;
        tl[j] = (j < k) ? y[ij] : x[ij];
    }
    __syncthreads();
    BP(1);
    // boundary rows: one wave per pivot column, coalesced along rows; four columns per pass so that their loads
    // and butterflies overlap (a wave owns up to k / 16 columns, each a dependent load -> reduce chain)
    // A front with a handful of boundary rows (the root: the border row alone) takes one thread per column instead: the
    // butterflies of 511 columns for one row each kept the LDS pipeline of the workgroup busy for 17 us.
    if (m - k <= 8) {
        for (int q = tid; q < k; q += nt) {
            const double* Aq = Fm + (int64_t)q * m;
            double sq = 0.0;
            for (int r = k; r < m; ++r) sq += Aq[r] * tl[r];
            gl[q] = sq;
        }
    } else
    for (int q0 = 4 * wave; q0 < k; q0 += 4 * (nt / 64)) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int r = k + lane; r < m; r += 64) {
            const double t = tl[r];
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += Fm[(int64_t)min(q0 + u, k - 1) * m + r] * t;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {           // rows of 16 lanes on the data-parallel path, the four rows through the crossbar
            s[u] += quad_perm_f64<0xB1>(s[u]);
            s[u] += quad_perm_f64<0x4E>(s[u]);
            s[u] += quad_perm_f64<0x124>(s[u]);
            s[u] += quad_perm_f64<0x128>(s[u]);
            s[u] += __shfl_xor(s[u], 16, 64);
            s[u] += __shfl_xor(s[u], 32, 64);
        }
        if (lane < 4 && q0 + lane < k) gl[q0 + lane] = s[lane == 0 ? 0 : (lane == 1 ? 1 : (lane == 2 ? 2 : 3))];
    }
    BP(2);
    const int wa = tid % NB, wb = tid / NB;
    const int last = ((k - 1) / NB) * NB;
    {
        const double w0 = (wa < wb && last + wb < k) ? Fm[(last + wa) + (int64_t)(last + wb) * m] : 0.0;
        if (wa < wb) { Ml[wb * (NB + 1) + wa] = w0; Ml[wa * (NB + 1) + wb] = w0; }
        if (tid < NB) Ml[tid * (NB + 1) + tid] = (last + tid < k) ? dv[last + tid] : 0.0;
    }
    __syncthreads();
    // Panel rows of a block step: G[q] += sum_u A[j0 + u, q] x[j0 + u] for every unsolved pivot column q < j0.  Column q
    // holds its 32 entries contiguously (256 B), so FOUR lanes share a column: per load instruction they cover 64
    // contiguous bytes (16-byte loads, 8-byte aligned) and a wave touches 16 cache lines instead of 64 -- one lane per
    // column made the texture addresser the bottleneck (6 us per step on the 511-pivot root front).  The four partial
    // sums meet in two butterfly steps, in a fixed order.
    struct __attribute__((aligned(8))) D2 { double a, b; };
    const int cq = tid >> 2, cp = tid & 3;           // column within a pass of nt / 4 columns, quarter of the column
    constexpr int CPP = BIGI_THREADS / 4;
    // The first two passes of a step's panel rows are requested at the top of the step and run under the block
    // product.  A column's base address is formed once.
    const double* col0 = Fm + (int64_t)min(cq, k - 1) * m + 2 * cp;
    const double* col1 = Fm + (int64_t)min(cq + CPP, k - 1) * m + 2 * cp;
    BP(3);
    for (int j0 = last; j0 >= 0; j0 -= NB) {
        const int nb = min(NB, k - j0);
        const int jn = j0 - NB;              // the next block is a full one
        if (j0 == 256) BP(4);
        if (j0 == 224) BP(8);
        const double wnext = (wa < wb && jn >= 0) ? Fm[(jn + wa) + (int64_t)(jn + wb) * m] : 0.0;
        const double dnext = (tid < NB && jn >= 0) ? dv[jn + tid] : 0.0;
        D2 pa[2][4];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const double* Aq = (ps ? col1 : col0) + j0;
            if (cq + ps * CPP < j0) {
                if (nb == NB) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) pa[ps][t] = *reinterpret_cast<const D2*>(Aq + 8 * t);
                } else {             // only the first step of a sweep can be a partial block
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int o = 2 * cp + 8 * t;
                        pa[ps][t].a = Aq[min(o, nb - 1) - 2 * cp];
                        pa[ps][t].b = Aq[min(o + 1, nb - 1) - 2 * cp];
                    }
                }
            }
        }
        {   // x_j = u_j - M_j G_j with all 1024 threads (see the forward sweep)
            const int q = tid >> 5, c = tid & 31;
            double ph = Ml[q * (NB + 1) + c] * (c < nb ? gl[j0 + c] : 0.0);
            ph += quad_perm_f64<0xB1>(ph);          // quads, then rotations by 4 and 8 inside the row of 16 lanes (DPP) ...
            ph += quad_perm_f64<0x4E>(ph);
            ph += quad_perm_f64<0x124>(ph);
            ph += quad_perm_f64<0x128>(ph);
            ph += __shfl_xor(ph, 16, 32);           // ... and one exchange between the two rows through the LDS crossbar
            if (c == 0) zq[q] = (q < nb) ? tl[j0 + q] - ph : 0.0;
        }
        if (j0 == 256) BP(5);
        __syncthreads();
        if (j0 == 256) BP(6);
        if (tid < nb) x[il[j0 + tid]] = zq[tid];
        if (wa < wb) { Ml[wb * (NB + 1) + wa] = wnext; Ml[wa * (NB + 1) + wb] = wnext; }
        if (tid < NB) Ml[tid * (NB + 1) + tid] = dnext;
        double zr[8];                        // this lane's eight entries of x_j (zero beyond nb)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            zr[2 * t] = zq[2 * cp + 8 * t];
            zr[2 * t + 1] = zq[2 * cp + 8 * t + 1];
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int q = cq + ps * CPP;
            if (q < j0) {                    // the four lanes of a column decide alike
                double v = 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) v += pa[ps][t].a * zr[2 * t] + pa[ps][t].b * zr[2 * t + 1];
                v += quad_perm_f64<0xB1>(v);
                v += quad_perm_f64<0x4E>(v);
                if (cp == 0) gl[q] += v;
            }
        }
        for (int q = cq + 2 * CPP; q - cq < j0; q += CPP) {          // fronts with more than 512 unsolved columns
            const double* Aq = Fm + (int64_t)min(q, j0 - 1) * m + j0 + 2 * cp;
            D2 a[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (nb == NB) a[t] = *reinterpret_cast<const D2*>(Aq + 8 * t);
                else {
                    const int o = 2 * cp + 8 * t;
                    a[t].a = Aq[min(o, nb - 1) - 2 * cp];
                    a[t].b = Aq[min(o + 1, nb - 1) - 2 * cp];
                }
            }
            double v = 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) v += a[t].a * zr[2 * t] + a[t].b * zr[2 * t + 1];
            v += quad_perm_f64<0xB1>(v);
            v += quad_perm_f64<0x4E>(v);
            if (cp == 0 && q < j0) gl[q] += v;
        }
        if (j0 == 256) BP(7);
        __syncthreads();
    }
    BP(9);
}

}  // namespace

void MfSolver::analyze(int64_t n, const int32_t* rowptr, const int32_t* colidx, hipStream_t st,
                       const double* coords, int dim, bool protect_peeled, const int32_t* top, int64_t ntop) {
    MfOptions opt;
    opt.protect_peeled = protect_peeled;
    opt.top = top;
    opt.ntop = ntop;
    if (const char* e = getenv("MGBHIP_NO_GEO"); e && e[0] == '1') coords = nullptr;
    opt.border = true;          // every system is factored bordered (mf_analysis.hpp): the Newton solve needs no forward sweep
    mf_analyze(n, rowptr, colidx, opt, plan, coords, dim);
    const int32_t nf = (int32_t)plan.fronts.size();
    std::vector<FrontDev> fd(nf);
    for (int32_t i = 0; i < nf; ++i) {
        const Front& f = plan.fronts[i];
        fd[i] = FrontDev{f.k, f.m, f.nchild, f.a_cnt, f.F_off, f.idx_off, f.u_off, f.child_off, f.rel_off, f.a_off, f.acol_off, -1, 0, 0};
    }
    d_front_idx.upload(plan.front_idx, st);
    d_children.upload(plan.children, st);
    d_rel.upload(plan.rel, st);
    d_a_src.upload(plan.a_src, st);
    d_a_colptr.upload(plan.a_colptr, st);
    d_arena.alloc((size_t)std::max<int64_t>(plan.arena_doubles, 1));
    d_uvec.alloc((size_t)std::max<int64_t>(plan.uvec_doubles, 1));
    d_y.alloc((size_t)plan.n + 1);            // + the border unknown
    y_zero = y_border_one = status_zero = leaf_zero = false;
    y_border_one = false;
    d_bx.alloc((size_t)plan.n + 1);
    d_xx.alloc((size_t)plan.n + 1);
    {
        const double one = 1.0;
        d_one.upload(&one, 1, st);
        MGB_HIP_CHECK(hipStreamSynchronize(st));
    }
    d_tbig.alloc(plan.front_idx.size() ? plan.front_idx.size() : 1);
    d_tsol.alloc(plan.front_idx.size() ? plan.front_idx.size() : 1);
    d_dvec.alloc(plan.front_idx.size() ? plan.front_idx.size() : 1);
    d_status.alloc(2);           // [0] factorization, [1] leaf pivots of a condensing f2
    d_status.zero(st);

    // dynamic LDS above 64 KB needs an explicit opt-in; fall back to the 64 KB classes if refused
    lds_cap = 88;
    if (hipFuncSetAttribute((const void*)mf_factor_small<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (128 * 128 + 16 * 128) * 8) == hipSuccess)
        lds_cap = 128;
    else
        (void)hipGetLastError();

    // inverse-based large-front path: needs the > 64 KB dynamic LDS opt-in for its single-workgroup solves
    bool inv_ok = true;
    if (const char* e = getenv("MGBHIP_OLD_BIG"); e && e[0] == '1') inv_ok = false;
    if (inv_ok) {
        const int lds = (2 * BIG_INV_MAX_M + BIG_INV_MAX_M / 2 + NB * (NB + 1) + 4 * NB + 8 + BIGI_THREADS) * (int)sizeof(double);    // 157 728 B of the 160 KB
        if (hipFuncSetAttribute((const void*)mf_fwd_inv, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)mf_bwd_inv, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            (void)hipGetLastError();
            inv_ok = false;
        }
    }
    static const int32_t classes[] = {16, 32, 48, 64, 88, 128};
    // Size gates of the two fastest kernel families (A/B switches of tests/test_gpu_solver.py).  Round 2 kept both off
    // systems of < 1024 unknowns after two creeping solves failed with them.  Round 3: all kernel selections are
    // equally backward stable on graded matrices (2.9e-13 componentwise) and the 27-problem sweep agrees with the
    // oracle without any gate (the 37-unknown case that motivated the wave gate: 5356 vs 5355 iterations), so the
    // one-wave kernel is ungated.  The inverse-based large-front path keeps its gate: without it config 4's phase I
    // (fem3d L=6, 9 000 iterations hugging the wall on a 145-unknown level) ends in "Initial centering failed" --
    // applying W = L_jj^{-1} is only forward stable in cond(L_jj), and such systems gain nothing from it.
    static const int64_t inv_min_n = [] { const char* e = getenv("MGBHIP_INV_MIN_N"); return e ? atoll(e) : 1024ll; }();
    static const bool merge_groups = [] { const char* e = getenv("MGBHIP_NO_MERGE_GROUPS"); return !(e && e[0] == '1'); }();
    static const int64_t wave_min_n = [] { const char* e = getenv("MGBHIP_WAVE_MIN_N"); return e ? atoll(e) : 0ll; }();
    uses_inv = false;
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
            L.tiny = (l == 0 && cls == 16);     // leaves with m <= 16: 16 lanes per front
            L.inv = (cls == 0 && L.max_m <= BIG_INV_MAX_M && inv_ok && plan.n >= inv_min_n);
            uses_inv = uses_inv || L.inv;
            for (int32_t q = i; q < j; ++q) {
                L.max_k = std::max(L.max_k, plan.fronts[q].k);
                L.max_child = std::max(L.max_child, plan.fronts[q].nchild);
            }
            level_launches[l].push_back(L);
            i = j;
        }
        // Launches of one level are independent but share a stream: a straggler group (4 fronts of the next
        // smaller class, 60 LDS-sized fronts beside 196 large ones) costs a full, latency-bound launch of
        // 20-30 us.  Fold small groups into their neighbour:
        //   (1) LDS-class fronts of a level whose bulk is on the large-front path join that path (it handles any m);
        //   (2) an LDS class with few fronts joins the next larger LDS class of the level.
        // Fronts are sorted by m inside a level, so a merge just extends the neighbour's range downwards.
        if (merge_groups) {
            auto& G = level_launches[l];
            auto absorb = [&](size_t into, size_t from) {        // from == into - 1
                MfLaunch& A = G[into];
                const MfLaunch& B = G[from];
                A.first = B.first;
                A.count += B.count;
                A.max_k = std::max(A.max_k, B.max_k);
                A.max_child = std::max(A.max_child, B.max_child);
                G.erase(G.begin() + (long)from);
            };
            if (G.size() >= 2 && G.back().cls == 0 && G.back().inv) {
                int64_t lds_count = 0;
                bool ok = true;
                for (size_t g = 0; g + 1 < G.size(); ++g) { lds_count += G[g].count; ok = ok && !G[g].tiny && plan.fronts[G[g].first].m > 32; }
                if (ok && lds_count <= G.back().count)
                    while (G.size() >= 2) absorb(G.size() - 1, G.size() - 2);
            }
            for (size_t g = 0; g + 1 < G.size();) {
                const bool next_lds = G[g + 1].cls != 0;
                if (next_lds && !G[g].tiny && (G[g].count < 256 || 4 * (int64_t)G[g].count < G[g + 1].count)) absorb(g + 1, g);
                else ++g;
            }
        }
    }
    if (plan.iface_front >= 0) {
        // the interface front gets a launch of its own on the large-front path (assemble / reduce / factor are separate
        // kernels there, whatever its size)
        const int32_t q = plan.iface_front;
        for (auto& G : level_launches) {
            for (size_t g = 0; g < G.size(); ++g) {
                MfLaunch& L = G[g];
                if (q < L.first || q >= L.first + L.count) continue;
                auto part = [&](int32_t first, int32_t count) {
                    MfLaunch P = L;
                    P.first = first; P.count = count;
                    P.max_m = 0; P.max_k = 0; P.max_child = 0;
                    for (int32_t t = first; t < first + count; ++t) {
                        P.max_m = std::max(P.max_m, plan.fronts[t].m);
                        P.max_k = std::max(P.max_k, plan.fronts[t].k);
                        P.max_child = std::max(P.max_child, plan.fronts[t].nchild);
                    }
                    return P;
                };
                std::vector<MfLaunch> out;
                if (q > L.first) out.push_back(part(L.first, q - L.first));
                MfLaunch I = part(q, 1);
                I.cls = 0; I.tiny = false; I.wave = false; I.iface = true;
                I.inv = (I.max_m <= BIG_INV_MAX_M && inv_ok && plan.n >= inv_min_n);
                uses_inv = uses_inv || I.inv;
                out.push_back(I);
                if (q + 1 < L.first + L.count) out.push_back(part(q + 1, L.first + L.count - q - 1));
                G.erase(G.begin() + (long)g);
                G.insert(G.begin() + (long)g, out.begin(), out.end());
                break;
            }
        }
    }
    std::vector<char> on_big_path((size_t)nf, 0);       // fronts of the large-front launches (after the merges above)
    for (auto& lev : level_launches)
        for (auto& L : lev)
            if (!L.cls)
                for (int32_t q = L.first; q < L.first + L.count; ++q) on_big_path[q] = 1;
    // update-vector gather lists of the large fronts (forward solve): for every local index the entries of
    // the children's update vectors that land on it, in child order (the summation order of the extend-add)
    std::vector<int64_t> ug_ptr, ug_src;
    for (int32_t i = 0; i < nf; ++i) {
        const Front& f = plan.fronts[i];
        if (!on_big_path[i]) continue;
        fd[i].ug_off = (int64_t)ug_ptr.size();
        std::vector<int32_t> cnt((size_t)f.m + 1, 0);
        for (int32_t c = 0; c < f.nchild; ++c) {
            const Front& ch = plan.fronts[plan.children[f.child_off + c]];
            for (int32_t j = 0; j < ch.m - ch.k; ++j) cnt[plan.rel[ch.rel_off + j] + 1]++;
        }
        const int64_t base = (int64_t)ug_src.size();
        std::vector<int64_t> pos((size_t)f.m + 1);
        pos[0] = base;
        for (int32_t j = 0; j < f.m; ++j) pos[j + 1] = pos[j] + cnt[j + 1];
        ug_ptr.insert(ug_ptr.end(), pos.begin(), pos.end());
        ug_src.resize((size_t)pos[f.m]);
        std::vector<int64_t> fill(pos.begin(), pos.end() - 1);
        for (int32_t c = 0; c < f.nchild; ++c) {
            const Front& ch = plan.fronts[plan.children[f.child_off + c]];
            for (int32_t j = 0; j < ch.m - ch.k; ++j) ug_src[(size_t)fill[plan.rel[ch.rel_off + j]]++] = ch.u_off + j;
        }
    }
    {   // Leaf fronts (m <= 16, the 16-lanes-per-front kernels) as packed lower triangles: m(m+1)/2 contiguous doubles
        // instead of m*m, read back by their parents' extend-add and by the sweeps.  Only when every parent is an LDS
        // front (the large-front assembly kernels read square children).
        static const bool no_pack = [] { const char* e = getenv("MGBHIP_NO_PACKED_LEAVES"); return e && e[0] == '1'; }();
        bool ok = !no_pack;
        for (auto& lev : level_launches)
            for (auto& L : lev)
                if (L.tiny)
                    for (int32_t q = L.first; q < L.first + L.count && ok; ++q) {
                        const int32_t par = plan.fronts[q].parent;
                        ok = par < 0 || !on_big_path[par];
                    }
        leaf_packed = false;
        if (ok)
            for (auto& lev : level_launches)
                for (auto& L : lev)
                    if (L.tiny) {
                        leaf_packed = true;
                        for (int32_t q = L.first; q < L.first + L.count; ++q) fd[q].packed = 1;
                    }
    }
    if (ug_ptr.empty()) ug_ptr.push_back(0);
    if (ug_src.empty()) ug_src.push_back(0);
    d_ug_ptr.upload(ug_ptr, st);
    d_ug_src.upload(ug_src, st);
    d_fronts.upload(fd, st);
    h_fronts = fd;
    {   // Packed LDS triangles.  Leaf fronts with m <= 16 (mf_factor_tiny) scatter A with column stride 16: 136 LDS
        // doubles per front instead of 256, which doubles the resident workgroups of that kernel.  Launches of fronts
        // with m <= 48 whose children are all small (update block <= 8 x 8: the element leaves under a level-1
        // front, or no children) go to the one-wave-per-front kernel mf_factor_wave, packed with stride 32 or 48;
        // with large children the 256-thread kernel's extend-add is faster and the launch stays there.
        std::vector<int32_t> ad(plan.a_dst);
        static const bool no_wave = [] { const char* e = getenv("MGBHIP_NO_WAVE_SMALL"); return e && e[0] == '1'; }();
        for (int32_t l = 0; l < nlev; ++l)
            for (auto& L : level_launches[l]) {
                int stride = 0;
                if (L.tiny) stride = 16;
                else if (L.cls && L.cls <= 48 && !no_wave && plan.n >= wave_min_n) {
                    bool ok = true;
                    for (int32_t q = L.first; q < L.first + L.count && ok; ++q) {
                        const Front& f = plan.fronts[q];
                        for (int32_t c = 0; c < f.nchild && ok; ++c) {
                            const Front& ch = plan.fronts[plan.children[f.child_off + c]];
                            ok = (ch.m - ch.k) * (ch.m - ch.k) <= 64;
                        }
                    }
                    L.wave = ok;
                    if (ok) stride = L.cls <= 32 ? 32 : 48;
                }
                const bool lds_front = L.cls >= 88 && !stride;  // mf_factor_small, packed classes: the front's own m as stride
                if (!stride && !lds_front) continue;
                for (int32_t q = L.first; q < L.first + L.count; ++q) {
                    const Front& f = plan.fronts[q];
                    const int32_t sd = lds_front ? f.m : stride;
                    for (int32_t t = 0; t < f.a_cnt; ++t) {
                        const int32_t d = plan.a_dst[f.a_off + t], lu = d % f.m, lv = d / f.m;      // row lu >= column lv
                        ad[f.a_off + t] = lv * sd - lv * (lv - 1) / 2 + (lu - lv);
                    }
                }
            }
        d_a_dst.upload(ad, st);
        MGB_HIP_CHECK(hipStreamSynchronize(st));
        h_a_dst.swap(ad);
    }
    // the wave-per-front solve kernels do not depend on the LDS class: one launch per level
    level_solves.assign(nlev, {});
    for (int32_t l = 0; l < nlev; ++l) {
        MfLaunch S{};
        for (auto& L : level_launches[l]) {
            if (!L.cls || L.tiny) { level_solves[l].push_back(L); continue; }
            if (S.count == 0) S = L;
            else {
                S.count += L.count;
                S.max_m = std::max(S.max_m, L.max_m);
                S.max_k = std::max(S.max_k, L.max_k);
            }
        }
        if (S.count) level_solves[l].insert(level_solves[l].begin(), S);
    }
    int32_t max_big = 1;
    for (auto& lev : level_launches)
        for (auto& L : lev)
            if (!L.cls) max_big = std::max(max_big, L.count);
    d_dscr.alloc((size_t)max_big * 2 * NB * NB);
    if (const char* e = getenv("MGBHIP_DEBUG"); e && atoi(e) >= 2) {
        fprintf(stderr, "[mgbhip] solver plan: n=%lld fronts=%d levels=%d arena=%.1f MB\n", (long long)plan.n, nf, nlev,
                plan.arena_doubles * 8e-6);
        for (int32_t l = 0; l < nlev; ++l)
            for (auto& L : level_launches[l]) {
                double sm = 0, sk = 0, fl = 0;
                for (int32_t q = L.first; q < L.first + L.count; ++q) {
                    const Front& f = plan.fronts[q];
                    sm += f.m; sk += f.k;
                    for (int c = 0; c < f.k; ++c) fl += (double)(f.m - c) * (f.m - c);
                }
                fprintf(stderr, "[mgbhip]   level %2d cls %3d count %7d max_m %4d max_k %4d avg_m %6.1f avg_k %6.1f Mflop %8.2f\n",
                        l, L.cls, L.count, L.max_m, L.max_k, sm / L.count, sk / L.count, fl * 1e-6);
            }
    }
    analyzed = true;
    MGB_HIP_CHECK(hipStreamSynchronize(st));   // host staging vectors go out of scope
}


// per-tree-level stage timers (MGBHIP_LEVEL_TIMING=1): "fac_lvNN", "fwd_lvNN", "bwd_lvNN"
static StageTimers g_dummy_timers;
static StageTimers* timers_or_dummy(StageTimers* t, bool on) { return (t && on) ? t : &g_dummy_timers; }

bool MfSolver::launch_big_assemble(const MfLaunch& L, dim3 ga, const double* d_values, const int32_t* a_src_p, hipStream_t st,
                                   bool with_diag) {
    const size_t lds = (size_t)L.max_child * (size_t)L.max_m * sizeof(int32_t);
    if (L.max_child >= 1 && L.max_child <= GATHER_MAX_CHILD && lds <= 40 * 1024) {
        const int ct = CT;                 // (one column per wave on levels with few fronts, ct = 4: no gain, measured in round 4)
        if (with_diag) ga.x += 1;          // the diagonal-block workgroup
        hipLaunchKernelGGL(mf_big_gather, ga, dim3(256), lds, st, cur_fr, L.first, d_children.p, d_rel.p, a_src_p,
                           cur_adst, cur_acol, d_values, d_arena.p, L.max_m, d_dscr.p, d_status.p, with_diag ? 1 : 0, ct);
        return with_diag;
    } else
        hipLaunchKernelGGL(mf_big_assemble, ga, dim3(256), 0, st, cur_fr, L.first, d_children.p, d_rel.p, a_src_p,
                           cur_adst, cur_acol, d_values, d_arena.p);
    return false;
}

void MfSolver::set_direct_map(const int32_t* value_map, int64_t nnz, int64_t tail_base, hipStream_t st) {
    std::vector<int32_t> as(plan.a_src.size());
    for (size_t t = 0; t < as.size(); ++t) {
        const int64_t src = plan.a_src[t];
        const int64_t v = src < nnz ? (int64_t)value_map[src] : tail_base + (src - nnz);
        MGB_REQUIRE(v < (int64_t)INT32_MAX, "direct value index exceeds 32 bits");
        as[t] = (int32_t)v;
    }
    d_a_src_direct.upload(as, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
}

void MfSolver::factor(const double* d_values, hipStream_t st, StageTimers* timers, bool direct, bool condensed) {
    MGB_REQUIRE(!direct || d_a_src_direct.n > 0, "MfSolver::factor: no direct value map");
    MGB_REQUIRE(!condensed || (direct && condensed_ok), "MfSolver::factor: condensed leaves are not enabled");
    // condensed: the leaf fronts were written by the element kernel (kernels.hpp, launch_elem_f2_condense); the other
    // fronts take only their border entries from the value array -- every element contribution reaches them through
    // the leaves' update blocks
    const int32_t* a_src_p = condensed ? d_a_src_c.p : (direct ? d_a_src_direct.p : d_a_src.p);
    cur_fr = condensed ? d_fronts_c.p : d_fronts.p;
    cur_adst = condensed ? d_a_dst_c.p : d_a_dst.p;
    cur_acol = condensed ? d_a_colptr_c.p : d_a_colptr.p;
    MGB_REQUIRE(analyzed, "MfSolver::factor before analyze");
    if (timers) timers->begin("factor");
    factored_inv = !robust;
    if (!status_zero) MGB_HIP_CHECK(hipMemsetAsync(d_status.p, 0, sizeof(int32_t), st));      // [1], the leaf flag of a condensing f2, stays
    status_zero = false;
    factored_condensed = condensed;
    static const bool lvl_timing = [] { const char* e = getenv("MGBHIP_LEVEL_TIMING"); return e && e[0] == '1'; }();
    int lvno = -1;
    for (auto& lev : level_launches) {
        ++lvno;
        char nm[32];
        snprintf(nm, sizeof(nm), "fac_lv%02d", lvno);
        StageScope lvscope(*timers_or_dummy(timers, lvl_timing), nm);
        for (auto& L : lev) {
            if (L.count == 0) continue;
            if (condensed && lvno == 0) continue;          // written by the element kernel
            if (L.tiny) {
                hipLaunchKernelGGL(mf_factor_tiny, dim3((L.count + 15) / 16), dim3(256), 0, st, cur_fr, L.first,
                                   L.count, a_src_p, cur_adst, d_values, d_arena.p, d_status.p);
            } else if (L.wave) {
                const dim3 gw((L.count + 3) / 4);
                if (L.cls <= 32) {
                    const size_t lds = (size_t)4 * (32 * 33 / 2) * sizeof(double) + 4 * 64 * (sizeof(int64_t) + sizeof(int32_t));
                    hipLaunchKernelGGL(mf_factor_wave<32>, gw, dim3(256), lds, st, cur_fr, L.first, L.count, d_children.p, d_rel.p,
                                       a_src_p, cur_adst, d_values, d_arena.p, d_status.p);
                } else {
                    const size_t lds = (size_t)4 * (48 * 49 / 2) * sizeof(double) + 4 * 64 * (sizeof(int64_t) + sizeof(int32_t));
                    hipLaunchKernelGGL(mf_factor_wave<48>, gw, dim3(256), lds, st, cur_fr, L.first, L.count, d_children.p, d_rel.p,
                                       a_src_p, cur_adst, d_values, d_arena.p, d_status.p);
                }
            } else if (L.cls) {
                // 8-column LDS panels (16- and 32-column ones were measured slower in round 3 and removed in round 4)
                const int threads = L.cls <= 16 ? 64 : (L.cls <= 32 ? 128 : 256);
                const bool packed = L.cls >= 88;
                const size_t lds = (size_t)((packed ? L.cls * (L.cls + 1) / 2 : L.cls * L.cls) + 8 * L.cls) * sizeof(double);    // front + scaled panel
                if (packed)
                    hipLaunchKernelGGL((mf_factor_small<8, true>), dim3(L.count), dim3(threads), lds, st, cur_fr, L.first, d_children.p,
                                       d_rel.p, a_src_p, cur_adst, d_values, d_arena.p, d_status.p);
                else
                    hipLaunchKernelGGL((mf_factor_small<8, false>), dim3(L.count), dim3(threads), lds, st, cur_fr, L.first, d_children.p,
                                       d_rel.p, a_src_p, cur_adst, d_values, d_arena.p, d_status.p);
            } else if (L.iface) {
                // assemble this rank's contribution, sum over ranks, then factor the complete front (every rank the same)
                MGB_REQUIRE((bool)iface_reduce, "MfSolver: interface front without a reduction hook");
                const Front& fi = plan.fronts[L.first];
                const dim3 ga((L.max_m + CT - 1) / CT, 1);
                launch_big_assemble(L, ga, d_values, a_src_p, st, false);
                {   // sum the lower triangle over ranks: (m + 1) m / 2 doubles instead of m^2
                    const int64_t tri = (int64_t)fi.m * (fi.m + 1) / 2;
                    d_ifpack.ensure((size_t)tri);
                    hipLaunchKernelGGL(mf_tri_pack, dim3(fi.m), dim3(256), 0, st, fi.m, d_arena.p + fi.F_off, d_ifpack.p, 0, (double*)nullptr);
                    iface_reduce(d_ifpack.p, tri);
                    hipLaunchKernelGGL(mf_tri_pack, dim3(fi.m), dim3(256), 0, st, fi.m, (const double*)nullptr, d_ifpack.p, 1, d_arena.p + fi.F_off);
                }
                if (L.inv && !robust) {
                    hipLaunchKernelGGL(mf_big_diag0, dim3(1), dim3(256), 0, st, cur_fr, L.first, d_arena.p, d_dscr.p, d_status.p);
                    for (int j0 = 0; j0 < L.max_k; j0 += NB) {
                        const int rem = L.max_m - j0;
                        const int T = std::max(0, (rem - 1 + ST - 1) / ST);
                        hipLaunchKernelGGL(mf_big_step, dim3(T * (T + 1) / 2 + 1, 1), dim3(256), 0, st, cur_fr, L.first, j0, d_arena.p,
                                           d_dscr.p, d_dvec.p, d_status.p, 0);
                    }
                } else {
                    for (int j0 = 0; j0 < L.max_k; j0 += NB) {
                        const int rem = L.max_m - j0;
                        hipLaunchKernelGGL(mf_big_panel, dim3(std::max(1, (rem - 1 + TR - 1) / TR), 1), dim3(256), 0, st, cur_fr,
                                           L.first, j0, d_arena.p, d_dscr.p, d_status.p, j0 == 0 ? 1 : 0);
                        const int T = (rem - 1 + ST - 1) / ST;
                        if (T > 0)
                            hipLaunchKernelGGL(mf_big_update, dim3(T * (T + 1) / 2 + 1, 1), dim3(256), 0, st, cur_fr, L.first, j0,
                                               d_arena.p, d_dscr.p, d_status.p);
                    }
                }
            } else if (L.inv && !robust) {
                const dim3 ga((L.max_m + CT - 1) / CT, L.count);
                // block 0 of every front is factored by an extra workgroup of the gather launch when that kernel applies;
                // otherwise once per front up front (many fronts) or inside every tile of step 0 (few fronts)
                const bool diag_done = launch_big_assemble(L, ga, d_values, a_src_p, st, true);
                const bool pre_diag = diag_done || L.count >= 24;
                if (pre_diag && !diag_done)
                    hipLaunchKernelGGL(mf_big_diag0, dim3(L.count), dim3(256), 0, st, cur_fr, L.first, d_arena.p, d_dscr.p,
                                       d_status.p);
                for (int j0 = 0; j0 < L.max_k; j0 += NB) {
                    const int rem = L.max_m - j0;
                    const int T = std::max(0, (rem - 1 + ST - 1) / ST);
                    const dim3 gs(T * (T + 1) / 2 + 1, L.count);         // trailing tiles + the look-ahead workgroup
                    hipLaunchKernelGGL(mf_big_step, gs, dim3(256), 0, st, cur_fr, L.first, j0, d_arena.p, d_dscr.p,
                                       d_dvec.p, d_status.p, (j0 == 0 && !pre_diag) ? 1 : 0);
                }
            } else {
                const dim3 ga((L.max_m + CT - 1) / CT, L.count);
                launch_big_assemble(L, ga, d_values, a_src_p, st, false);
                for (int j0 = 0; j0 < L.max_k; j0 += NB) {
                    const int rem = L.max_m - j0;                // rows from the panel start, at most
                    const dim3 gp(std::max(1, (rem - 1 + TR - 1) / TR), L.count);
                    hipLaunchKernelGGL(mf_big_panel, gp, dim3(256), 0, st, cur_fr, L.first, j0, d_arena.p,
                                       d_dscr.p, d_status.p, j0 == 0 ? 1 : 0);
                    const int T = (rem - 1 + ST - 1) / ST;       // trailing tiles (upper bound)
                    if (T > 0) {
                        const dim3 gu(T * (T + 1) / 2 + 1, L.count);     // + the look-ahead workgroup
                        hipLaunchKernelGGL(mf_big_update, gu, dim3(256), 0, st, cur_fr, L.first, j0, d_arena.p,
                                           d_dscr.p, d_status.p);
                    }
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
    // the factored system is [H c; c' gamma] with c = 0 (set_border_identity): solve it for [b; 0]
    const size_t n = (size_t)plan.n;
    MGB_HIP_CHECK(hipMemcpyAsync(d_bx.p, d_b, n * sizeof(double), hipMemcpyDeviceToDevice, st));
    MGB_HIP_CHECK(hipMemsetAsync(d_bx.p + n, 0, sizeof(double), st));
    forward_pass(d_bx.p, st, timers);
    backward_pass(d_xx.p, st, timers);
    MGB_HIP_CHECK(hipMemcpyAsync(d_x, d_xx.p, n * sizeof(double), hipMemcpyDeviceToDevice, st));
    MGB_HIP_CHECK(hipGetLastError());
    if (timers) timers->end();
}

void MfSolver::solve_border(double* d_x_np1, hipStream_t st, StageTimers* timers) {
    MGB_REQUIRE(analyzed, "MfSolver::solve_border before analyze");
    if (timers) timers->begin("trisolve");
    // factors of [H -g; -g' -1]: the forward substitution of H x = g already ran as the border row of every
    // front.  L' x = e_n backwards from x_n = 1 gives x[0:n] = H^{-1} g.
    const size_t n = (size_t)plan.n;
    if (!y_zero) {                   // the backward sweeps only read y: it stays zero from one Newton iteration to the next
        MGB_HIP_CHECK(hipMemsetAsync(d_y.p, 0, n * sizeof(double), st));
        y_zero = true;
    }
    if (!y_border_one) {             // y[n] = 1 survives the backward sweeps: set once (a generic forward sweep overwrites it)
        MGB_HIP_CHECK(hipMemcpyAsync(d_y.p + n, d_one.p, sizeof(double), hipMemcpyDeviceToDevice, st));
        y_border_one = true;
    }
    backward_pass(d_x_np1, st, timers);
    MGB_HIP_CHECK(hipGetLastError());
    if (timers) timers->end();
}

void MfSolver::forward_pass(const double* d_b, hipStream_t st, StageTimers* timers) {
    y_border_one = false;
    y_zero = false;
    static const bool lvl_timing = [] { const char* e = getenv("MGBHIP_LEVEL_TIMING"); return e && e[0] == '1'; }();
    int lvno = -1;
    for (auto& lev : level_solves) {
        ++lvno;
        char nm[32];
        snprintf(nm, sizeof(nm), "fwd_lv%02d", lvno);
        StageScope lvscope(*timers_or_dummy(timers, lvl_timing), nm);
        for (auto& L : lev) {
            if (L.count == 0) continue;
            if (L.tiny) {
                hipLaunchKernelGGL(mf_forward_tiny, dim3((L.count + 15) / 16), dim3(256), 0, st, d_fronts.p, L.first,
                                   L.count, d_front_idx.p, d_arena.p, d_b, d_y.p, d_uvec.p);
            } else if (L.cls) {
                const int ts = (L.max_m + 1) & ~1;
                hipLaunchKernelGGL(mf_forward_small, dim3((L.count + 3) / 4), dim3(256), (size_t)4 * ts * sizeof(double),
                                   st, d_fronts.p, L.first, L.count, ts, d_front_idx.p, d_children.p, d_rel.p,
                                   d_arena.p, d_b, d_y.p, d_uvec.p);
            } else if (L.inv && factored_inv) {
                const size_t lds = (size_t)(((L.max_m + 1) & ~1) + NB * (NB + 1) + 2 * NB + BIGI_THREADS) * sizeof(double);
                hipLaunchKernelGGL(mf_fwd_inv, dim3(L.count), dim3(BIGI_THREADS), lds, st, d_fronts.p, L.first,
                                   d_front_idx.p, d_ug_ptr.p, d_ug_src.p, d_arena.p, d_dvec.p, d_b, d_y.p, d_uvec.p);
            } else if (L.max_m <= BIG1_MAX_M) {
                const size_t lds = (size_t)(((L.max_m + 1) & ~1) + NB * (NB + 1)) * sizeof(double);
                hipLaunchKernelGGL(mf_fwd_big1, dim3(L.count), dim3(BIG1_THREADS), lds, st, d_fronts.p, L.first,
                                   d_front_idx.p, d_children.p, d_rel.p, d_arena.p, d_b, d_y.p, d_uvec.p);
            } else {
                const dim3 gi((L.max_m + 255) / 256, L.count);
                hipLaunchKernelGGL(mf_fwd_big_init, gi, dim3(256), 0, st, d_fronts.p, L.first, d_front_idx.p,
                                   d_children.p, d_rel.p, d_b, d_uvec.p, d_tbig.p);
                for (int j0 = 0; j0 < L.max_k; j0 += NB) {
                    const int rem = L.max_m - j0;
                    const dim3 gs(std::max(1, (rem - 1 + 255) / 256), L.count);
                    hipLaunchKernelGGL(mf_fwd_big_step, gs, dim3(256), 0, st, d_fronts.p, L.first, j0, d_arena.p,
                                       d_tbig.p, d_tsol.p);
                }
                hipLaunchKernelGGL(mf_fwd_big_fin, gi, dim3(256), 0, st, d_fronts.p, L.first, d_front_idx.p,
                                   d_arena.p, d_tbig.p, d_tsol.p, d_y.p, d_uvec.p);
            }
        }
    }
}

void MfSolver::backward_pass(double* d_x, hipStream_t st, StageTimers* timers) {
    static const bool lvl_timing = [] { const char* e = getenv("MGBHIP_LEVEL_TIMING"); return e && e[0] == '1'; }();
    for (int32_t l = (int32_t)level_solves.size() - 1; l >= 0; --l) {
        char nm[32];
        snprintf(nm, sizeof(nm), "bwd_lv%02d", l);
        StageScope lvscope(*timers_or_dummy(timers, lvl_timing), nm);
        for (auto it = level_solves[l].rbegin(); it != level_solves[l].rend(); ++it) {
            const MfLaunch& L = *it;
            if (L.count == 0) continue;
            if (L.tiny) {
                hipLaunchKernelGGL(mf_backward_tiny, dim3((L.count + 15) / 16), dim3(256), 0, st, d_fronts.p, L.first,
                                   L.count, d_front_idx.p, d_arena.p, d_y.p, d_x);
            } else if (L.cls) {
#define MGB_LAUNCH_BWD_SMALL(KM)                                                                                          \
    hipLaunchKernelGGL(mf_backward_small<KM>, dim3((L.count + 3) / 4), dim3(256), 0, st, d_fronts.p, L.first, L.count, \
                       d_front_idx.p, d_arena.p, d_y.p, d_x)
                // (a 32-column variant holds 143 registers and loses more to occupancy on the 8192-front level than it gains)
                if (L.max_k <= 8) MGB_LAUNCH_BWD_SMALL(8);
                else if (L.max_k <= 16) MGB_LAUNCH_BWD_SMALL(16);
                else MGB_LAUNCH_BWD_SMALL(0);
#undef MGB_LAUNCH_BWD_SMALL
            } else if (L.inv && factored_inv) {
                const size_t lds = (size_t)(((L.max_m + 1) & ~1) + ((L.max_k + 1) & ~1) + NB * (NB + 1) + 2 * NB + (L.max_m + 1) / 2) * sizeof(double);
                hipLaunchKernelGGL(mf_bwd_inv, dim3(L.count), dim3(BIGI_THREADS), lds, st, d_fronts.p, L.first,
                                   d_front_idx.p, d_arena.p, d_dvec.p, d_y.p, d_x);
            } else if (L.max_m <= BIG1_MAX_M) {
                const size_t lds = (size_t)(((L.max_m + 1) & ~1) + NB * (NB + 1)) * sizeof(double);
                hipLaunchKernelGGL(mf_bwd_big1, dim3(L.count), dim3(BIG1_THREADS), lds, st, d_fronts.p, L.first,
                                   d_front_idx.p, d_arena.p, d_y.p, d_x);
            } else {
                const dim3 gi((L.max_k + 3) / 4, L.count);
                hipLaunchKernelGGL(mf_bwd_big_init, gi, dim3(256), 0, st, d_fronts.p, L.first, d_front_idx.p,
                                   d_arena.p, d_y.p, d_x, d_tbig.p);
                const int last = ((L.max_k - 1) / NB) * NB;
                for (int j0 = last; j0 >= 0; j0 -= NB) {
                    const dim3 gs(std::max(1, (j0 + 255) / 256), L.count);
                    hipLaunchKernelGGL(mf_bwd_big_step, gs, dim3(256), 0, st, d_fronts.p, L.first, j0, d_front_idx.p,
                                       d_arena.p, d_tbig.p, d_x);
                }
            }
        }
    }
}

#ifdef MGB_STEP_PROBE
void mf_debug_probe(long long* out64) {
    (void)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_probe), 64 * sizeof(long long));
    long long init[64];
    for (int i = 0; i < 64; ++i) init[i] = 0;
    init[56] = 0x7fffffffffffffffll;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_probe), init, sizeof(init));
}
#endif

void MfSolver::chain_stats(double* out) const {
    double blocks = 0, big_levels = 0, fac = 0, bwd = 0, extra = 0;
    for (auto& lev : level_launches) {
        int lev_blocks = 0;
        for (auto& L : lev) {
            if (L.count == 0) continue;
            if (L.tiny || L.wave || L.cls) { fac += 1; continue; }
            const int nb = (L.max_k + NB - 1) / NB;
            lev_blocks = std::max(lev_blocks, nb);
            fac += 1 + nb + (L.iface ? 3 : 0);
            for (int32_t q = L.first; q < L.first + L.count; ++q) {
                const Front& f = plan.fronts[q];
                const double mk = (double)(f.m - f.k), steps = (double)((f.k + NB - 1) / NB);
                // every 32-column step reads and writes the trailing lower triangle it updates; one pass is the floor
                extra += 2.0 * 0.5 * (mk * mk + mk * f.k) * std::max(0.0, steps - 1.0);
            }
        }
        if (lev_blocks) { big_levels += 1; blocks += lev_blocks; }
    }
    for (auto& lev : level_solves)
        for (auto& L : lev)
            if (L.count) bwd += 1;
    out[0] = blocks; out[1] = big_levels; out[2] = fac; out[3] = bwd;
    out[4] = (double)plan.arena_doubles; out[5] = (double)plan.factor_flops; out[6] = extra; out[7] = 0.0;
}

void MfSolver::status_async(int32_t* h_dst2, hipStream_t st) const {
    MGB_HIP_CHECK(hipMemcpyAsync(h_dst2, d_status.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
}

int MfSolver::status(hipStream_t st) {
    int32_t h[2] = {0, 0};
    d_status.download(h, 2, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    return status_from(h, factored_condensed);
}

// ---- condensed leaves ----------------------------------------------------------------------------------------
bool MfSolver::enable_condensed(int64_t N, int P, const int32_t* ucol, int64_t slack0, int64_t nnz, int64_t tail_base,
                                hipStream_t st) {
    condensed_ok = false;
    int why = 0;
    struct Report { int& w; ~Report() { if (w) if (const char* e = getenv("MGBHIP_DEBUG"); e && atoi(e) >= 2) fprintf(stderr, "[mgbhip] condensed leaves: plan shape check %d failed\n", w); } } report{why};
    const int32_t nlev = (int32_t)plan.level_ptr.size() - 1;
    if (!analyzed || !plan.border || nlev < 2 || P > 7) { why = 1; return false; }
    const int32_t n0 = plan.level_ptr[1];
    if ((int64_t)n0 != N) { why = 2; return false; }
    const int32_t nbor = (int32_t)plan.n;
    std::vector<LeafDesc> desc((size_t)N);
    std::vector<char> seen((size_t)N, 0);
    for (int32_t i = 0; i < n0; ++i) {
        const Front& f = plan.fronts[i];
        const int32_t* idx = plan.front_idx.data() + f.idx_off;
        if (f.k != P + 1 || f.m > 15 || f.nchild != 0 || idx[f.m - 1] != nbor) { why = 3; return false; }
        const int64_t e = ((int64_t)idx[0] - slack0) / P;
        if (e < 0 || e >= N || seen[(size_t)e]) { why = 4; return false; }
        for (int q = 0; q < P; ++q)
            if ((int64_t)idx[q] != slack0 + e * P + q) { why = 5; return false; }           // pivots: the element's slacks in node order ...
        if (idx[P] != ucol[e * P + (P - 1)] || idx[P] < 0) { why = 6; return false; }       // ... then its interior u node
        uint32_t packed = (uint32_t)f.m;
        int found = 0;
        for (int q = 0; q < P - 1; ++q) {
            uint32_t pos = 15;
            const int32_t c = ucol[e * P + q];
            if (c >= 0)
                for (int32_t t = P + 1; t < f.m - 1; ++t)
                    if (idx[t] == c) { pos = (uint32_t)t; ++found; break; }
            packed |= pos << (4 + 4 * q);
        }
        if (found != f.m - 1 - (P + 1)) { why = 7; return false; }                           // every boundary unknown is a node of the element
        for (int q = 0; q < P - 1; ++q)                                          // ... and every node that is an unknown is on the boundary
            if (ucol[e * P + q] >= 0 && ((packed >> (4 + 4 * q)) & 15u) == 15u) { why = 8; return false; }
        packed |= 15u << (4 + 4 * (P - 1));
        desc[(size_t)e] = LeafDesc{f.F_off, idx[P], packed};
        seen[(size_t)e] = 1;
    }
    // A lists of the other fronts without the matrix entries (which arrive through the leaves): the border column only
    const int32_t nf = (int32_t)plan.fronts.size();
    std::vector<FrontDev> fd(h_fronts);
    std::vector<int32_t> as, ad, ac;
    for (int32_t i = 0; i < nf; ++i) {
        const Front& f = plan.fronts[i];
        fd[i].a_off = (int64_t)as.size();
        fd[i].acol_off = (int64_t)ac.size();
        const int32_t* cp = plan.a_colptr.data() + f.acol_off;
        for (int32_t c = 0; c < f.k; ++c) {
            ac.push_back((int32_t)((int64_t)as.size() - fd[i].a_off));
            if (i < n0) continue;
            for (int32_t t = cp[c]; t < cp[c + 1]; ++t) {
                const int64_t src = plan.a_src[f.a_off + t];
                if (src < nnz) continue;
                const int64_t v = tail_base + (src - nnz);
                if (v >= (int64_t)INT32_MAX) return false;
                as.push_back((int32_t)v);
                ad.push_back(h_a_dst[f.a_off + t]);
            }
        }
        fd[i].a_cnt = (int32_t)((int64_t)as.size() - fd[i].a_off);
        ac.push_back(fd[i].a_cnt);
    }
    if (as.empty()) { as.push_back(0); ad.push_back(0); }
    d_fronts_c.upload(fd, st);
    d_a_src_c.upload(as, st);
    d_a_dst_c.upload(ad, st);
    d_a_colptr_c.upload(ac, st);
    d_leaf_desc.upload(desc, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    condensed_ok = true;
    return true;
}

}  // namespace mgbhip
