// 32 x 32 LDL' of a diagonal block on ONE wave with the trailing updates on the fp64 matrix cores.
// Included by mf_numeric.hip (inside its namespace, after fast_recip / readlane_f64) and by tools/micro/ldlt32_mfma_test.hip.
#pragma once

struct Blk4 { double w10, w20, w21, w30, w31, w32, i0, i1, i2, i3, d0, d1, d2, d3; };

__device__ __forceinline__ void ldlt4_core(double a00, double a10, double a11, double a20, double a21, double a22, double a30,
                                           double a31, double a32, double a33, Blk4& B, double (&l)[6]) {
    const double d0 = a00, i0 = fast_recip(d0);
    const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
    const double d1 = a11 - l10 * a10, i1 = fast_recip(d1);
    const double t21 = a21 - l20 * a10, t31 = a31 - l30 * a10;
    const double l21 = t21 * i1, l31 = t31 * i1;
    const double d2 = a22 - l20 * a20 - l21 * t21, i2 = fast_recip(d2);
    const double t32 = a32 - l30 * a20 - l31 * t21;
    const double l32 = t32 * i2;
    const double d3 = a33 - l30 * a30 - l31 * t31 - l32 * t32, i3 = fast_recip(d3);
    B.d0 = d0; B.d1 = d1; B.d2 = d2; B.d3 = d3;
    B.i0 = i0; B.i1 = i1; B.i2 = i2; B.i3 = i3;
    B.w10 = -l10; B.w21 = -l21; B.w32 = -l32;
    B.w20 = l21 * l10 - l20;
    B.w31 = l32 * l21 - l31;
    B.w30 = l31 * l10 + l32 * (l20 - l21 * l10) - l30;
    l[0] = l10; l[1] = l20; l[2] = l21; l[3] = l30; l[4] = l31; l[5] = l32;
}
__device__ __forceinline__ bool ldlt4_serial(double a00, double a10, double a11, double a20, double a21, double a22, double a30,
                                             double a31, double a32, double a33, Blk4& B, double (&l)[6]) {
    ldlt4_core(a00, a10, a11, a20, a21, a22, a30, a31, a32, a33, B, l);
    const double dmin = fmin(fmin(fabs(B.d0), fabs(B.d1)), fmin(fabs(B.d2), fabs(B.d3)));
    return !(dmin > 0.0) || !isfinite(B.d0) || !isfinite(B.d1) || !isfinite(B.d2) || !isfinite(B.d3);
}

// The lower triangle lives in three 16 x 16 accumulator tiles of v_mfma_f64_16x16x4 (c00: rows/cols 0-15, c10: rows 16-31 x
// cols 0-15, c11: rows/cols 16-31): lane (fr = lane & 15, fk = lane >> 4) holds entry (row fr, col 4 i + fk) of a tile in
// register i -- so the FOUR columns of block step b (i = b & 3) already sit in the operand layout of the instruction
// (operand index k = fk), and a step is: ten diagonal-block entries by v_readlane -> the 4 x 4 LDL' redundantly in every
// lane -> the row's four entries through the crossbar -> s = a W4', l = s D^-1 -> C -= S L' as one MFMA per tile.  No LDS
// round trip, no workgroup barrier inside the 32 columns (the 4 x 4-blocked workgroup version needed two per four columns:
// 1 900 cycles per step, 6.4 us per block).
// Contract (same as block_ldlt32_b4): Dn holds the lower triangle incl. diagonal of the nb x nb block, row-major, zeros above
// the diagonal and beyond nb; on return the strictly lower unit factor (zeros elsewhere), dq the pivots (1 beyond nb).
// Every thread of the 256-thread workgroup must call this.
typedef double ld_double4_t __attribute__((ext_vector_type(4)));
// WITH_INV: the unit-lower inverse W = L^{-1} is built alongside, one row block behind the factorization, by wave 1
// (W[b,:] = W4_b (E_b - L[b,<b] W[<b,:]): a block forward substitution whose inputs -- the rows of L left of the diagonal
// block, final one step earlier, and W4_b, a by-product of the 4 x 4 step -- wave 0 publishes through LDS and a step
// counter; wave 0 never waits).  Wv receives W with unit diagonal and zero upper triangle, ~0.3 us after the last pivot
// (the recursive-doubling inverse that ran after the factorization took 1.6 us and seven workgroup barriers).
template <bool WITH_INV>
__device__ __forceinline__ void block_ldlt32_mfma_t(double (*Dn)[32 + 1], double* dq, int nb, int tid, double (*Wv)[32 + 1],
                                                    int32_t* __restrict__ status) {
    constexpr int N32 = 32;
    __shared__ double w4s[N32 / 4][6];
    __shared__ int ldl_step;
    if (tid < N32 && tid >= nb) Dn[tid][tid] = 1.0;      // identity padding keeps the recurrences free of special cases
    if (WITH_INV) {
        for (int i = tid; i < N32 * N32; i += 256) Wv[i / N32][i % N32] = (i / N32 == i % N32) ? 1.0 : 0.0;
        if (tid == 0) ldl_step = 0;
    }
    __syncthreads();
    if (WITH_INV && tid >= 64 && tid < 128) {
        const int c = tid & 31, half = (tid >> 5) & 1;   // one column of W per lane pair; the two half-waves split the k range
        for (int b = 0; b < N32 / 4; ++b) {
            while (__atomic_load_n(&ldl_step, __ATOMIC_RELAXED) < b + 1) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int cb = 4 * b;
            const double w10 = w4s[b][0], w20 = w4s[b][1], w21 = w4s[b][2], w30 = w4s[b][3], w31 = w4s[b][4], w32 = w4s[b][5];
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
            for (int k0 = 4 * half; k0 < cb; k0 += 8) {      // W[k][c] = 0 for k < c: the uniform trip count is exact
                double wk[4], la[4], lb[4], lc[4], ld[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {                 // twenty LDS reads in flight (a rolled loop paid a latency per k)
                    wk[u] = Wv[k0 + u][c];
                    la[u] = Dn[cb][k0 + u]; lb[u] = Dn[cb + 1][k0 + u]; lc[u] = Dn[cb + 2][k0 + u]; ld[u] = Dn[cb + 3][k0 + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { t0 -= la[u] * wk[u]; t1 -= lb[u] * wk[u]; t2 -= lc[u] * wk[u]; t3 -= ld[u] * wk[u]; }
            }
            t0 += __shfl_xor(t0, 32, 64); t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64); t3 += __shfl_xor(t3, 32, 64);
            if (half == 0) {
                if (c < cb) {
                    Wv[cb][c] = t0;
                    Wv[cb + 1][c] = t1 + w10 * t0;
                    Wv[cb + 2][c] = t2 + w20 * t0 + w21 * t1;
                    Wv[cb + 3][c] = t3 + w30 * t0 + w31 * t1 + w32 * t2;
                } else if (c < cb + 4) {                      // the diagonal block of W is W4_b itself
                    const int cc = c - cb;
                    if (cc == 0) { Wv[cb + 1][c] = w10; Wv[cb + 2][c] = w20; Wv[cb + 3][c] = w30; }
                    else if (cc == 1) { Wv[cb + 2][c] = w21; Wv[cb + 3][c] = w31; }
                    else if (cc == 2) { Wv[cb + 3][c] = w32; }
                }
            }
        }
    }
    if (tid < 64) {
        const int fr = tid & 15, fk = tid >> 4;
        ld_double4_t c00, c10, c11;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c00[i] = Dn[fr][4 * i + fk];
            c10[i] = Dn[16 + fr][4 * i + fk];
            c11[i] = Dn[16 + fr][16 + 4 * i + fk];
        }
        double dmin = 1.0, dnan = 0.0;             // smallest |pivot| of the block; a non-finite pivot turns dnan into NaN
#pragma unroll
        for (int b = 0; b < N32 / 4; ++b) {
            const int T = b >> 2, i = b & 3, cb = 4 * b, cn = cb + 4;
            const double src = T == 0 ? c00[i] : c11[i];
            // the rows' four panel entries through the crossbar first: their latency runs under the 4 x 4 recurrence
            double pa[2][4];
            {
                const double xa = T == 0 ? c00[i] : c11[i], xb = c10[i];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pa[0][q] = __shfl(xa, fr + 16 * q, 64);
                    pa[1][q] = (T == 0) ? __shfl(xb, fr + 16 * q, 64) : 0.0;
                }
            }
            const int l0 = 4 * i;                        // lane of entry (cb, cb); (cb + rr, cb + cc) is lane l0 + rr + 16 cc
            const double a00 = readlane_f64(src, l0);
            const double a10 = readlane_f64(src, l0 + 1), a11 = readlane_f64(src, l0 + 1 + 16);
            const double a20 = readlane_f64(src, l0 + 2), a21 = readlane_f64(src, l0 + 2 + 16), a22 = readlane_f64(src, l0 + 2 + 32);
            const double a30 = readlane_f64(src, l0 + 3), a31 = readlane_f64(src, l0 + 3 + 16), a32 = readlane_f64(src, l0 + 3 + 32),
                         a33 = readlane_f64(src, l0 + 3 + 48);
            Blk4 B;
            double l[6];
            ldlt4_core(a00, a10, a11, a20, a21, a22, a30, a31, a32, a33, B, l);
            dmin = fmin(fmin(dmin, fabs(B.d0)), fmin(fabs(B.d1), fmin(fabs(B.d2), fabs(B.d3))));
            dnan = fma(B.d0, 0.0, fma(B.d1, 0.0, fma(B.d2, 0.0, fma(B.d3, 0.0, dnan))));
            // panel rows of one tile row from the gathered entries: the operands s(row)[fk], l(row)[fk] of the update
            const double iv = fk == 0 ? B.i0 : (fk == 1 ? B.i1 : (fk == 2 ? B.i2 : B.i3));
            auto panel = [&](const double (&a)[4], int rowbase, double& ms, double& ml) {
                const double s0 = a[0];
                const double s1 = a[1] + a[0] * B.w10;
                const double s2 = a[2] + a[0] * B.w20 + a[1] * B.w21;
                const double s3 = a[3] + a[0] * B.w30 + a[1] * B.w31 + a[2] * B.w32;
                const double sv = fk == 0 ? s0 : (fk == 1 ? s1 : (fk == 2 ? s2 : s3));
                const double lv = sv * iv;
                const int row = rowbase + fr;
                const bool valid = row >= cn;             // rows above the block are finished, rows inside it come from the 4 x 4
                ms = valid ? sv : 0.0;
                ml = valid ? lv : 0.0;
                if (valid) Dn[row][cb + fk] = lv;
            };
            if (T == 0) {
                double sA, lA, sB, lB;
                panel(pa[0], 0, sA, lA);
                panel(pa[1], 16, sB, lB);
                if (cn < 16) c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(lA, -sA, c00, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(lA, -sB, c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(lB, -sB, c11, 0, 0, 0);
            } else if (cn < N32) {
                double sB, lB;
                panel(pa[0], 16, sB, lB);
                c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(lB, -sB, c11, 0, 0, 0);
            }
            // pivots and the block's own multipliers (and, WITH_INV, the hand-off to the inverse's wave): after the update is issued
            if (tid == 0) {
                dq[cb] = B.d0; dq[cb + 1] = B.d1; dq[cb + 2] = B.d2; dq[cb + 3] = B.d3;
                Dn[cb + 1][cb] = l[0]; Dn[cb + 2][cb] = l[1]; Dn[cb + 2][cb + 1] = l[2];
                Dn[cb + 3][cb] = l[3]; Dn[cb + 3][cb + 1] = l[4]; Dn[cb + 3][cb + 2] = l[5];
                if (WITH_INV) {
                    w4s[b][0] = B.w10; w4s[b][1] = B.w20; w4s[b][2] = B.w21; w4s[b][3] = B.w30; w4s[b][4] = B.w31; w4s[b][5] = B.w32;
                    // rows of L left of block b were final after step b - 1 (this wave's LDS stores stay ordered)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __atomic_store_n(&ldl_step, b + 1, __ATOMIC_RELAXED);
                }
            }
        }
        const bool bad = !(dmin > 0.0) || !(dnan == 0.0);      // same rule as ldlt4_serial: a zero or non-finite pivot
        if (bad && status && tid == 0) atomicOr(status, 1);
    }
    __syncthreads();
    // strictly lower L only: clear the diagonal and everything above it
    for (int i = tid; i < N32 * N32; i += 256) {
        const int rr = i / N32, cc = i % N32;
        if (cc >= rr || rr >= nb) Dn[rr][cc] = 0.0;
    }
    __syncthreads();
}
__device__ __forceinline__ void block_ldlt32_mfma(double (*Dn)[32 + 1], double* dq, int nb, int tid, int32_t* __restrict__ status) {
    block_ldlt32_mfma_t<false>(Dn, dq, nb, tid, nullptr, status);
}
__device__ __forceinline__ void block_ldlt32_inv_mfma(double (*Dn)[32 + 1], double* dq, int nb, int tid, double (*Wv)[32 + 1],
                                                      int32_t* __restrict__ status) {
    block_ldlt32_mfma_t<true>(Dn, dq, nb, tid, Wv, status);
}

// W = L^{-1} for the unit lower triangular 32 x 32 L in Ls (strictly lower part, row-major) by recursive doubling
//   W21 = -W22 (L21 W11)     for blocks of 4 -> 8 -> 16 -> 32 rows
// on ONE wave with every product on the matrix cores: the pairs of a level are packed into one 16 x 16 (x K) MFMA with
// block-diagonal operands -- lane (fr, fk) feeds A[m = fr][k = fk] and B[k = fk][n = fr] and keeps the result registers
// whose block (m >> log2 s) matches its column's (n >> log2 s).  Operands come from LDS, results go back through it;
// the only synchronisation inside is the wave's own LDS ordering (the 256-thread form needed seven workgroup barriers:
// 1.6 us per block on the critical path of the pivot chain).  Wv receives W with unit diagonal and a zero upper triangle.
// Every thread of the workgroup must call this (two barriers).
__device__ __forceinline__ void ld_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void block_inverse32_mfma(const double (*Ls)[32 + 1], double (*Wv)[32 + 1], double (*Tm)[17], int tid) {
    constexpr int N32 = 32;
    for (int i = tid; i < N32 * N32; i += 256) Wv[i / N32][i % N32] = (i / N32 == i % N32) ? 1.0 : 0.0;
    __syncthreads();
    if (tid < 64) {
        const int fr = tid & 15, fk = tid >> 4;
        const ld_double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
        if (tid < N32 / 4) {        // 4 x 4 diagonal blocks in closed form
            const int o = 4 * tid;
            const double l21 = Ls[o + 1][o], l31 = Ls[o + 2][o], l32 = Ls[o + 2][o + 1];
            const double l41 = Ls[o + 3][o], l42 = Ls[o + 3][o + 1], l43 = Ls[o + 3][o + 2];
            Wv[o + 1][o] = -l21;
            Wv[o + 2][o + 1] = -l32;
            Wv[o + 3][o + 2] = -l43;
            Wv[o + 2][o] = l32 * l21 - l31;
            Wv[o + 3][o + 1] = l43 * l32 - l42;
            Wv[o + 3][o] = l42 * l21 + l43 * (l31 - l32 * l21) - l41;
        }
        ld_wave_lds_sync();
        {   // s = 4: four pairs, rows 8p .. 8p+7; operand row m <-> (p = m >> 2, i = m & 3), column n <-> (p', j)
            const int p = fr >> 2, i = fr & 3;
            ld_double4_t t = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[8 * p + 4 + i][8 * p + fk], Wv[8 * p + fk][8 * p + i], zero4, 0, 0, 0);
            // D[m = fk + 4 r][n = fr]: block of m is r, i = fk; keep r == p (the column's block), j = fr & 3
            const double tv = p == 0 ? t[0] : (p == 1 ? t[1] : (p == 2 ? t[2] : t[3]));
            Tm[4 * p + fk][i] = tv;
            ld_wave_lds_sync();
            ld_double4_t w = __builtin_amdgcn_mfma_f64_16x16x4f64(Wv[8 * p + 4 + i][8 * p + 4 + fk], Tm[4 * p + fk][i], zero4, 0, 0, 0);
            const double wv = p == 0 ? w[0] : (p == 1 ? w[1] : (p == 2 ? w[2] : w[3]));
            Wv[8 * p + 4 + fk][8 * p + i] = -wv;
            ld_wave_lds_sync();
        }
        {   // s = 8: two pairs, rows 16p .. 16p+15; m <-> (p = m >> 3, i = m & 7), n <-> (p', j = n & 7)
            const int p = fr >> 3, i = fr & 7;
            ld_double4_t t = zero4;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                t = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[16 * p + 8 + i][16 * p + 4 * kb + fk], Wv[16 * p + 4 * kb + fk][16 * p + i], t, 0, 0, 0);
            // D[m = fk + 4 r][n = fr]: block of m is r >> 1, row in block 4 (r & 1) + fk; keep r >> 1 == p
            Tm[8 * p + fk][i] = p == 0 ? t[0] : t[2];
            Tm[8 * p + 4 + fk][i] = p == 0 ? t[1] : t[3];
            ld_wave_lds_sync();
            ld_double4_t w = zero4;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                w = __builtin_amdgcn_mfma_f64_16x16x4f64(Wv[16 * p + 8 + i][16 * p + 8 + 4 * kb + fk], Tm[8 * p + 4 * kb + fk][i], w, 0, 0, 0);
            Wv[16 * p + 8 + fk][16 * p + i] = -(p == 0 ? w[0] : w[2]);
            Wv[16 * p + 8 + 4 + fk][16 * p + i] = -(p == 0 ? w[1] : w[3]);
            ld_wave_lds_sync();
        }
        {   // s = 16: one pair
            ld_double4_t t = zero4;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                t = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[16 + fr][4 * kb + fk], Wv[4 * kb + fk][fr], t, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Tm[fk + 4 * r][fr] = t[r];
            ld_wave_lds_sync();
            ld_double4_t w = zero4;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                w = __builtin_amdgcn_mfma_f64_16x16x4f64(Wv[16 + fr][16 + 4 * kb + fk], Tm[4 * kb + fk][fr], w, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Wv[16 + fk + 4 * r][fr] = -w[r];
        }
    }
    __syncthreads();
}
