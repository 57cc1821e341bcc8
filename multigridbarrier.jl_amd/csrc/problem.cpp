// problem.cpp -- native_to_device for one (AMG, Convex) pair, the per-level R'HR assembly
// plans, and the device-resident f0/f1/f2/solve primitives behind the C ABI.
#include "problem.hpp"
#include "plan_device.hpp"
#include "dense.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <atomic>
#include <exception>
#include <cstdlib>
#include <string>
#include <thread>
#include <cstring>
#include <numeric>

using namespace mgbhip;

static constexpr int64_t ACC_MAX_M = 384;


// ---------------------------------------------------------------------------------------------
// problem construction
// ---------------------------------------------------------------------------------------------

static void upload_csr(const mgbhip_csr& R, Level& L, hipStream_t st) {
    MGB_REQUIRE(R.rows >= 0 && R.cols >= 0 && R.rowptr && (R.rowptr[R.rows] == 0 || (R.colidx && R.values)),
                "bad prolongation CSR");
    MGB_REQUIRE(R.rows < INT32_MAX && R.cols < INT32_MAX, "prolongation too large for 32-bit indices");
    L.rows = R.rows;
    L.m = R.cols;
    const int64_t nnz = R.rowptr[R.rows];
    L.hRptr.assign(R.rowptr, R.rowptr + R.rows + 1);
    L.hRcol.assign(R.colidx, R.colidx + nnz);
    L.hRval.assign(R.values, R.values + nnz);
    for (int64_t i = 0; i < R.rows; ++i) {
        MGB_REQUIRE(L.hRptr[i] <= L.hRptr[i + 1], "CSR row pointers must be non-decreasing");
        for (int32_t q = L.hRptr[i]; q < L.hRptr[i + 1]; ++q)
            MGB_REQUIRE(L.hRcol[q] >= 0 && L.hRcol[q] < R.cols, "CSR column index out of range");
    }
    L.Rptr.upload(L.hRptr, st);
    L.Rcol.upload(L.hRcol.data(), L.hRcol.size(), st);
    L.Rval.upload(L.hRval.data(), L.hRval.size(), st);
    if (L.hRcol.empty()) { L.Rcol.alloc(1); L.Rval.alloc(1); }
    // transpose (CSR of R') for the gather form of R' * v: on the device from the uploaded R (one stable radix sort by
    // column: rows ascending inside a column, as the host loop it replaced produced them) -- MGBHIP_HOST_TRANSPOSE=1 keeps the
    // host loop, which also sent the transposed arrays over PCIe (0.65 GB at L = 9)
    static const bool host_T = [] { const char* e = getenv("MGBHIP_HOST_TRANSPOSE"); return e && e[0] == '1'; }();
    std::vector<int32_t> tp, tc;
    std::vector<double> tv;
    int32_t maxrow = 0;
    if (host_T) {
        tp.assign((size_t)R.cols + 1, 0); tc.resize((size_t)nnz); tv.resize((size_t)nnz);
        for (int64_t q = 0; q < nnz; ++q) tp[L.hRcol[q] + 1]++;
        for (int64_t j = 0; j < R.cols; ++j) tp[j + 1] += tp[j];
        std::vector<int32_t> fill(tp.begin(), tp.end() - 1);
        for (int64_t i = 0; i < R.rows; ++i)
            for (int32_t q = L.hRptr[i]; q < L.hRptr[i + 1]; ++q) {
                int32_t d = fill[L.hRcol[q]]++;
                tc[d] = (int32_t)i;
                tv[d] = L.hRval[q];
            }
        for (int64_t j = 0; j < R.cols; ++j) maxrow = std::max(maxrow, tp[j + 1] - tp[j]);
    } else {
        maxrow = transpose_csr_device(R.rows, R.cols, nnz, L.Rptr.p, L.Rcol.p, L.Rval.p, L.Tptr, L.Tcol, L.Tval, st);
    }
    L.T_long = maxrow > 64;
    L.T_chunks = (maxrow >= 1024 && R.cols <= 16384) ? csr_chunks(maxrow) : 0;
    int32_t maxr = 0;
    for (int64_t i = 0; i < R.rows; ++i) maxr = std::max(maxr, L.hRptr[i + 1] - L.hRptr[i]);
    L.R_long = maxr > 64;
    {   // selection rows: the element kernels read s through the column map and no prolongation kernel runs
        bool unit = maxr <= 1 && getenv("MGBHIP_NO_FUSED_PROLONG") == nullptr;
        for (int64_t q = 0; q < nnz && unit; ++q) unit = L.hRval[q] == 1.0;
        L.R_unit = unit;
        if (unit) {
            std::vector<int32_t> sel((size_t)std::max<int64_t>(R.rows, 1), -1);
            for (int64_t i = 0; i < R.rows; ++i)
                if (L.hRptr[i + 1] > L.hRptr[i]) sel[(size_t)i] = L.hRcol[L.hRptr[i]];
            L.Rsel.upload(sel, st);
        }
    }
    if (host_T) {
        L.Tptr.upload(tp, st);
        L.Tcol.upload(tc.data(), tc.size(), st);
        L.Tval.upload(tv.data(), tv.size(), st);
        if (tc.empty()) { L.Tcol.alloc(1); L.Tval.alloc(1); }
    }
    MGB_HIP_CHECK(hipStreamSynchronize(st));
}

static const double* upload_grid(mgbhip_problem* P, const double* h, size_t count) {
    if (!h) return nullptr;
    P->cone_grids.emplace_back();
    P->cone_grids.back().upload(h, count, P->stream());
    return P->cone_grids.back().p;
}

mgbhip_problem* problem_create(mgbhip_ctx* ctx, const mgbhip_problem_desc* d, mgbhip_problem* share) {
    MGB_REQUIRE(ctx && d, "null argument");
    MGB_REQUIRE(d->p >= 1 && d->N >= 1, "need at least one element and one node per element");
    // p <= 64: element kernels.  p > 64 is accepted for ONE element only: dense spectral operators.
    MGB_REQUIRE(d->p <= 64 || d->N == 1, "more than 64 nodes per element is supported for a single dense element only");
    MGB_REQUIRE(d->p <= 16384, "dense element too large");
    MGB_REQUIRE(d->nu >= 1 && d->nu <= MGBHIP_MAX_NU, "state components out of range");
    MGB_REQUIRE(d->nD >= 1 && d->nD <= MGBHIP_MAX_ND, "D rows out of range");
    MGB_REQUIRE(d->n_ops >= 1 && d->n_ops <= MGBHIP_MAX_OPS, "operator count out of range");
    MGB_REQUIRE(d->L >= 1 && d->R && d->w, "missing hierarchy or weights");
    MGB_HIP_CHECK(hipSetDevice(ctx->device));
    const auto t_create0 = std::chrono::steady_clock::now();
    const bool dbg2 = [] { const char* e = getenv("MGBHIP_DEBUG"); return e && atoi(e) >= 2; }();
    auto since0 = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_create0).count(); };
    std::unique_ptr<mgbhip_problem> P(new mgbhip_problem());
    P->ctx = ctx;
    P->p = d->p;
    P->N = d->N;
    P->n = (int64_t)d->p * d->N;
    P->nu = d->nu;
    P->nD = d->nD;
    P->dense = d->p > 64;
    if (d->x && d->dim > 0 && d->dim <= 3 && !P->dense) {
        P->hx.assign(d->x, d->x + (size_t)d->dim * P->n);
        P->xdim = d->dim;
    }
    MGB_REQUIRE((int64_t)P->nu * P->n < INT32_MAX, "problem too large for 32-bit row indices");
    hipStream_t st = ctx->stream;
    const size_t blk = (size_t)d->p * d->p * d->N;
    if (share) {
        MGB_REQUIRE(share->ctx == ctx && share->p == P->p && share->N == P->N, "share: incompatible problem");
        MGB_REQUIRE((int)share->store->ops.size() >= d->n_ops, "share: operator list mismatch");
        P->store = share->store;
    } else {
        P->store = std::make_shared<OpStore>();
        P->store->ops.resize(d->n_ops);
        P->store->identity.resize(d->n_ops);
        for (int o = 0; o < d->n_ops; ++o) {
            P->store->identity[o] = d->ops[o] == nullptr;
            if (d->ops[o]) P->store->ops[o].upload(d->ops[o], blk, st);
        }
        P->store->w.upload(d->w, (size_t)P->n, st);
    }
    // D table + LDS staging decision
    P->nstage = 0;
    int slot_of_op[MGBHIP_MAX_OPS];
    for (int o = 0; o < MGBHIP_MAX_OPS; ++o) slot_of_op[o] = -1;
    const int G = elem_group(P->p);
    const int EPB = P->dense ? 1 : 256 / G;
    for (int k = 0; k < d->nD; ++k) {
        MGB_REQUIRE(d->D_state[k] >= 0 && d->D_state[k] < d->nu, "D row references a missing state variable");
        MGB_REQUIRE(d->D_op[k] >= 0 && d->D_op[k] < d->n_ops, "D row references a missing operator");
        P->D_state[k] = d->D_state[k];
        P->D_op[k] = d->D_op[k];
        const int o = d->D_op[k];
        if (P->store->identity[o]) { P->D_stage[k] = -1; continue; }
        if (slot_of_op[o] < 0) {
            // stage through LDS while the operator tiles of one workgroup stay under 64 KB
            // ... and the whole f2 working set (operators + broken values + nD(nD+1)/2 node weights
            // per lane) fits the 160 KB of a CU
            size_t bytes = (size_t)(P->nstage + 1) * EPB * P->p * P->p * sizeof(double);
            const size_t f2_total = bytes + 256 * sizeof(double) * (size_t)(d->nu + d->nD * (d->nD + 1) / 2);
            if (!P->dense && bytes <= 64 * 1024 && f2_total <= 150 * 1024) {
                slot_of_op[o] = P->nstage;
                P->stage_ptr[P->nstage++] = P->store->ops[o].p;
            } else {
                slot_of_op[o] = -2;
            }
        }
        P->D_stage[k] = slot_of_op[o];
    }
    for (int k = d->nD; k < MGBHIP_MAX_ND; ++k) { P->D_state[k] = -1; P->D_op[k] = 0; P->D_stage[k] = -1; }
    {
        bool sid[MGBHIP_MAX_NU];
        for (int a = 0; a < d->nu; ++a) {
            sid[a] = true;
            for (int k = 0; k < d->nD; ++k)
                if (d->D_state[k] == a && !P->store->identity[d->D_op[k]]) sid[a] = false;
        }
        P->diag_mask_sel = 0;
        for (int a = 0; a < d->nu; ++a)
            for (int b = a; b < d->nu; ++b)
                if (sid[a] && sid[b]) P->diag_mask_sel |= 1 << hel_block_index(a, b, d->nu);
    }

    // cone
    const mgbhip_cone& C = d->cone;
    MGB_REQUIRE(C.npieces >= 1 && C.npieces <= MGBHIP_MAX_PIECES, "unsupported number of convex pieces");
    std::memset(&P->cone, 0, sizeof(P->cone));
    P->cone.npieces = C.npieces;
    P->cone.feasibility = C.feasibility;
    P->cone.NC = C.NC;
    P->cone.box_b = 1.0;
    P->cone.box_R = 1.0;
    const int ny_piece = C.feasibility ? C.NC - 1 : d->nD;    // pieces index into the user D rows
    if (C.feasibility) MGB_REQUIRE(C.NC >= 2 && C.NC <= d->nD, "phase-I wrapper: bad NC");
    for (int k = 0; k < C.npieces; ++k) {
        const mgbhip_piece& s = C.pieces[k];
        PieceDev& t = P->cone.pc[k];
        MGB_REQUIRE(s.kind == MGBHIP_KIND_EP || s.kind == MGBHIP_KIND_LINEAR, "unknown functor family");
        MGB_REQUIRE(s.ni >= 1 && s.ni <= MGBHIP_MAX_IDX, "functor index list too long for this build");
        t.kind = s.kind;
        t.ni = s.ni;
        t.nc = (s.kind == MGBHIP_KIND_EP) ? s.ni : s.nc;
        MGB_REQUIRE(t.nc >= 1 && t.nc <= MGBHIP_MAX_IDX, "functor constraint count too large for this build");
        if (s.kind == MGBHIP_KIND_EP) MGB_REQUIRE(s.ni >= 2, "Euclidean power cone needs nz >= 2");
        for (int c = 0; c < MGBHIP_MAX_IDX; ++c) {
            t.idx[c] = c < s.ni ? s.idx[c] : 0;
            if (c < s.ni) MGB_REQUIRE(s.idx[c] >= 0 && s.idx[c] < ny_piece, "functor index outside the D rows");
        }
        if (!s.A) MGB_REQUIRE(t.nc == t.ni, "identity A needs a square constraint matrix");
        t.A = upload_grid(P.get(), s.A, (size_t)P->n * t.nc * t.ni);
        t.b = upload_grid(P.get(), s.b, (size_t)P->n * t.nc);
        t.p = upload_grid(P.get(), s.p, (size_t)P->n);
        t.mu = upload_grid(P.get(), s.mu, (size_t)P->n);
        t.select = upload_grid(P.get(), s.select, (size_t)P->n);
        t.p_const = s.p_const;
        t.mu_const = s.mu_const;
        if (s.kind == MGBHIP_KIND_EP && !s.p) MGB_REQUIRE(s.p_const >= 1.0, "power cone exponent must be >= 1");
    }
    if (d->barrier_weights) {
        P->bw.upload(d->barrier_weights, (size_t)P->n, st);
        P->has_bw = true;
    }
    if (dbg2) fprintf(stderr, "[mgbhip] problem_create: operators, weights and cone grids on the device after %.3f s\n", since0());
    // hierarchy
    P->levels.resize(d->L);
    for (int l = 0; l < d->L; ++l) MGB_REQUIRE(d->R[l].rows == (int64_t)P->nu * P->n, "prolongation row count must be nu*n");
    {   // the levels are independent and their upload is host work (validation, the CSR of R', selection maps): one host
        // thread per level, at most eight at a time (time to first solution: 0.6 s of single-thread loops at L = 9)
        std::vector<std::exception_ptr> errors((size_t)d->L);     // rethrown with their type (argument error vs HIP error)
        std::atomic<int> next{0};
        auto worker = [&] {
            (void)hipSetDevice(ctx->device);
            for (int l = next.fetch_add(1); l < d->L; l = next.fetch_add(1)) {
                try {
                    const double t0 = since0();
                    upload_csr(d->R[l], P->levels[l], st);
                    if (dbg2) fprintf(stderr, "[mgbhip] problem_create: level %d (nnz %lld) %.3f -> %.3f s\n", l,
                                      (long long)d->R[l].rowptr[d->R[l].rows], t0, since0());
                } catch (...) {
                    errors[(size_t)l] = std::current_exception();
                }
            }
        };
        const int nthreads = std::min(d->L, 8);
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
        for (const auto& e : errors)
            if (e) std::rethrow_exception(e);
    }
    if (dbg2) fprintf(stderr, "[mgbhip] problem_create: %d levels uploaded after %.3f s\n", d->L, since0());
    // workspace
    int64_t mmax = 1;
    for (auto& L : P->levels) mmax = std::max(mmax, L.m);
    const size_t zn = (size_t)P->nu * P->n;
    P->d_z.alloc(zn);
    P->d_z0.alloc(zn);
    P->d_zfull.alloc(zn);
    P->d_c.alloc((size_t)P->n * P->nD);
    P->d_c0.alloc((size_t)P->n * P->nD);
    P->d_ret.alloc(zn);
    if (P->dense) {
        P->d_dnDz.alloc((size_t)P->n * P->nD);
        P->d_dnY.alloc((size_t)P->n * (P->nD * (P->nD + 1) / 2));
        P->d_hel.alloc(1);
    } else {
        P->hel_cap = (int64_t)P->N * hel_blocks(P->nu) * P->p * P->p;
        P->d_hel.alloc((size_t)P->hel_cap);
    }
    P->d_partials.alloc((size_t)elem_grid(P->p, P->N));
    P->d_scal.alloc(16);
    P->d_scratch.alloc((size_t)reduce_scratch_doubles(std::max<int64_t>(mmax, (int64_t)zn)));
    P->d_nodeF.alloc((size_t)P->n);
    P->d_x.alloc(mmax); P->d_g.alloc(mmax); P->d_nv.alloc(mmax + 1); P->d_xn.alloc(mmax);   // d_nv: + the border entry of solve_border
    P->d_gn.alloc(mmax); P->d_tmp.alloc(std::max<size_t>(mmax, zn));   // d_tmp also holds the t-ramp's roll-back copy of z
    {
        size_t tch = 1;
        for (auto& L : P->levels) tch = std::max(tch, (size_t)L.T_chunks * (size_t)L.m);
        P->d_tchunk.alloc(2 * tch);            // (sum, carried error) per chunk
    }
    P->d_flag.alloc(4);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    if (dbg2) fprintf(stderr, "[mgbhip] problem_create: done after %.3f s\n", since0());
    return P.release();
}

// ---------------------------------------------------------------------------------------------
// assembly plan (reference: _make_block_assembly_plan, src/BlockMatrices.jl:322-491; the
// CUDA twin builds dense panels + Int32 scatter maps, block_ops.jl:251-411)
// ---------------------------------------------------------------------------------------------

// Dense spectral level: the Hessian is a full m x m matrix (its CSR image is the row-major
// array itself), H = DR' * (Ybar * DR) with DR_k = D_k * R[rows of state(k)] formed once here.
void mgbhip_problem::ensure_plan_dense(int level) {
    Level& L = levels[level];
    hipStream_t st = stream();
    const int64_t m = L.m, nn = n;
    MGB_REQUIRE(m * m < (int64_t)INT32_MAX, "dense Hessian exceeds 32-bit indexing");
    L.hHptr.resize(m + 1);
    L.hHcol.resize((size_t)(m * m));
    for (int64_t i = 0; i <= m; ++i) L.hHptr[i] = (int32_t)(i * m);
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < m; ++j) L.hHcol[i * m + j] = (int32_t)j;
    L.nnz = m * m;
    L.Hptr.upload(L.hHptr, st);
    L.Hcol.upload(L.hHcol, st);
    L.Hval.alloc((size_t)(L.nnz + L.m + 1));       // + the border column of the bordered factorization (mf_solver.hpp)
    L.selection = false;
    const int64_t ld = (int64_t)nD * nn;
    L.denseDR.alloc((size_t)std::max<int64_t>(ld * m, 1));
    L.denseW.alloc((size_t)std::max<int64_t>(ld * m, 1));
    MGB_HIP_CHECK(hipMemsetAsync(L.denseW.p, 0, sizeof(double) * (size_t)std::max<int64_t>(ld * m, 1), st));
    std::vector<DevBuf<double>> Rd(nu);          // dense R blocks per state (n x m, column-major)
    std::vector<double> host((size_t)(nn * m));
    DevBuf<double> opT;
    opT.alloc((size_t)(nn * nn));
    for (int k = 0; k < nD; ++k) {
        const int a = D_state[k];
        if (!Rd[a].p) {
            std::fill(host.begin(), host.end(), 0.0);
            for (int64_t r = 0; r < nn; ++r) {
                const int64_t row = (int64_t)a * nn + r;
                for (int32_t q = L.hRptr[row]; q < L.hRptr[row + 1]; ++q) host[(size_t)(r + nn * L.hRcol[q])] += L.hRval[q];
            }
            Rd[a].upload(host.data(), host.size(), st);
            MGB_HIP_CHECK(hipStreamSynchronize(st));
        }
        double* dst = L.denseDR.p + (int64_t)k * nn;
        const double* op = store->ops[D_op[k]].p;
        if (m == 0) continue;
        if (store->identity[D_op[k]]) {
            MGB_HIP_CHECK(hipMemcpy2DAsync(dst, sizeof(double) * ld, Rd[a].p, sizeof(double) * nn, sizeof(double) * nn,
                                           (size_t)m, hipMemcpyDeviceToDevice, st));
        } else {
            launch_dense_transpose((int)nn, op, opT.p, st);
            launch_dense_gemm_tn((int)nn, (int)m, (int)nn, opT.p, nn, nullptr, Rd[a].p, nn, dst, ld, false, false, st);
        }
    }
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    L.planned = true;
}

void mgbhip_problem::ensure_plan(int level) {
    Level& L = levels[level];
    if (L.planned) return;
    if (dense) { ensure_plan_dense(level); return; }
    const auto t_plan0 = std::chrono::steady_clock::now();
    struct PlanTimer {
        std::chrono::steady_clock::time_point t0; int level;
        ~PlanTimer() {
            if (const char* dbg = getenv("MGBHIP_DEBUG"); dbg && atoi(dbg) >= 2)
                fprintf(stderr, "[mgbhip] assembly plan level %d built in %.2f s\n", level,
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
    } plan_timer{t_plan0, level};
    hipStream_t st = stream();
    const int64_t NE = N;
    const int pp = p;
    const int64_t nn = n;
    // 1. per (element, state) column sets
    bool selection = true;
    for (size_t q = 0; q < L.hRval.size() && selection; ++q) selection = (L.hRval[q] == 1.0);
    for (int64_t i = 0; i < L.rows && selection; ++i) selection = (L.hRptr[i + 1] - L.hRptr[i] <= 1);
    std::vector<int32_t> ecol_ptr((size_t)NE * nu + 1, 0);
    std::vector<int32_t> ecols;
    ecols.reserve((size_t)NE * nu * pp);
    std::vector<int32_t> tmp;
    for (int64_t e = 0; e < NE; ++e)
        for (int a = 0; a < nu; ++a) {
            tmp.clear();
            for (int r = 0; r < pp; ++r) {
                const int64_t row = (int64_t)a * nn + e * pp + r;
                for (int32_t q = L.hRptr[row]; q < L.hRptr[row + 1]; ++q) tmp.push_back(L.hRcol[q]);
            }
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            ecols.insert(ecols.end(), tmp.begin(), tmp.end());
            ecol_ptr[e * nu + a + 1] = (int32_t)ecols.size();
            MGB_REQUIRE(ecols.size() < (size_t)INT32_MAX, "assembly plan exceeds 32-bit indexing");
        }
    // Blocks (a, b) whose D rows are all identity operators are diagonal per element
    // (Hel_ab[i, j] = sum_r delta_ri Y_r delta_rj): on selection levels their off-diagonal slots
    // are structural zeros, exactly as in the reference's sparse products, and stay out of the
    // pattern.  (With default_D this decouples the broken slack unknowns of an element.)
    bool state_id[MGBHIP_MAX_NU];
    for (int a = 0; a < nu; ++a) {
        state_id[a] = true;
        for (int k = 0; k < nD; ++k)
            if (D_state[k] == a && !store->identity[D_op[k]]) state_id[a] = false;
    }
    auto structural = [&](int a, int i, int b, int j) { return !(selection && state_id[a] && state_id[b] && i != j); };
    auto colof = [&](int a, int64_t e, int r) -> int32_t {
        const int64_t row = (int64_t)a * nn + e * pp + r;
        return L.hRptr[row + 1] > L.hRptr[row] ? L.hRcol[L.hRptr[row]] : -1;
    };
    // 2. output pattern
    const int64_t m = L.m;
    L.hHptr.assign(m + 1, 0);
    L.hHcol.clear();
    // Small general levels with wide supports: dense H through LDS accumulators (launch_panel_accumulate)
    int32_t cmax_all = 1;
    for (size_t q = 0; q + 1 < ecol_ptr.size(); ++q) cmax_all = std::max(cmax_all, ecol_ptr[q + 1] - ecol_ptr[q]);
    {
        int64_t slab_est = 0, ctmax = 1;
        for (int64_t e = 0; e < NE; ++e) {
            const int64_t ct = ecol_ptr[(e + 1) * nu] - ecol_ptr[e * nu];
            slab_est += ct * ct;
            ctmax = std::max(ctmax, ct);
        }
        L.acc_ctmax = (int32_t)ctmax;
        // LDS accumulate (launch_panel_accumulate): chunks of the packed upper triangle sized so that
        // chunk + one element's staging fit the LDS budget; element streams fill the GPU once
        const int64_t mt = m * (m + 1) / 2;
        const int64_t stage = (int64_t)panel_stage_doubles(pp, nu, (int)ctmax);
        const int64_t room = (int64_t)(PANEL_ACC_LDS_MAX / sizeof(double)) - stage;
        int64_t nsplit = room > 0 ? (mt + room - 1) / room : 1 << 20;
        if (nsplit < 1) nsplit = 1;
        const int64_t chunk = (mt + nsplit - 1) / nsplit;
        const int64_t waves = std::max<int64_t>(8, std::min<int64_t>(NE, (nsplit == 1 ? 512 : 256) / nsplit));
        L.acc_waves = (int32_t)waves;
        L.acc_split = (int32_t)nsplit;
        L.acc_chunk = (int32_t)chunk;
        // wide coarse supports only (3-D hierarchies): many contributions per entry, few enough
        // elements per stream; narrow supports are faster through slab + gather
        constexpr int64_t acc_ne_max = 65536;
        L.acc = !selection && m > 0 && m <= ACC_MAX_M && room > 0 && nsplit <= 4 && NE <= acc_ne_max && slab_est >= 16 * mt &&
                panel_accumulate_fits(pp, nu, (int)ctmax);
        if (const char* dbg = getenv("MGBHIP_DEBUG"); dbg && atoi(dbg) >= 2)
            fprintf(stderr, "[mgbhip] assembly plan level %d: m=%lld selection=%d cmax=%d slab=%lld doubles, accumulators=%lld -> %s\n", level,
                    (long long)m, (int)selection, cmax_all, (long long)slab_est, (long long)(waves * mt), L.acc ? "LDS accumulate" : "slab + gather");
    }
    // Pattern + contribution lists: one stable radix sort on the device (plan_device.hip) unless the pair
    // count is out of its range or MGBHIP_HOST_PLAN=1 asks for the host builder below (kept as the
    // cross-check: tests compare the two bit for bit).
    std::vector<int64_t> eoff;
    if (!selection && !L.acc) {
        eoff.assign((size_t)NE + 1, 0);
        for (int64_t e = 0; e < NE; ++e) {
            const int64_t ct = ecol_ptr[(e + 1) * nu] - ecol_ptr[e * nu];
            eoff[e + 1] = eoff[e] + ct * ct;
        }
        MGB_REQUIRE(eoff[NE] < (int64_t)INT32_MAX, "projected slab exceeds 32-bit indexing");
        L.slab_doubles = eoff[NE];
    }
    bool device_plan = false;
    if (!L.acc && m > 0) {
        static const bool host_plan = [] { const char* e = getenv("MGBHIP_HOST_PLAN"); return e && e[0] == '1'; }();
        PlanDeviceIn in;
        std::memset(&in, 0, sizeof(in));
        in.selection = selection; in.NE = NE; in.n = nn; in.m = m; in.p = pp; in.nu = nu;
        in.slab_doubles = L.slab_doubles;
        in.diag_mask_sel = diag_mask_sel;
        for (int a = 0; a < nu; ++a) if (state_id[a]) in.state_id_mask |= 1u << a;
        hel_layout(nu, NE, pp, diag_mask_sel, in.sel_off);
        static const bool no_direct = [] { const char* e = getenv("MGBHIP_NO_DIRECT"); return e && e[0] == '1'; }();
        in.extra_base = (selection && !no_direct && hel_cap > 0) ? hel_cap : 0;
        if (!host_plan && plan_device_pairs(in) <= PLAN_DEVICE_MAX_PAIRS) {
            if (selection) {
                in.Rptr = L.Rptr.p; in.Rcol = L.Rcol.p;
            } else {
                std::vector<int32_t> eoff32(eoff.begin(), eoff.end());
                L.eoff.upload(eoff32, st);
                L.ecol_ptr.upload(ecol_ptr, st);
                L.ecols.upload(ecols.data(), ecols.size(), st);
                if (ecols.empty()) L.ecols.alloc(1);
                MGB_HIP_CHECK(hipStreamSynchronize(st));     // eoff32 is a local
                in.ecol_ptr = L.ecol_ptr.p; in.ecols = L.ecols.p; in.eoff = L.eoff.p;
            }
            build_plan_device(in, L, st);
            device_plan = true;
            if (L.direct) {          // room for the shared sums and the border column behind the slab
                std::lock_guard<std::mutex> lock(shared_mutex);     // several selection levels may be planned at once (prepare_all)
                d_hel.ensure((size_t)(hel_cap + L.nshared + m + 1));
                hel_level = -1;
            }
        }
    }
    if (device_plan) {
        // pattern, lists and their host copies are in place
    } else if (L.acc) {
        L.hHcol.resize((size_t)(m * m));
        for (int64_t i = 0; i <= m; ++i) L.hHptr[i] = (int32_t)(i * m);
        for (int64_t i = 0; i < m; ++i)
            for (int64_t j = 0; j < m; ++j) L.hHcol[i * m + j] = (int32_t)j;
    } else if (selection) {
        // (row, col) pairs of every structural element contribution, bucketed by row
        std::vector<int32_t> rc(m + 1, 0);
        auto each_pair = [&](auto&& emit) {
            for (int64_t e = 0; e < NE; ++e)
                for (int a = 0; a < nu; ++a)
                    for (int i = 0; i < pp; ++i) {
                        const int32_t ci = colof(a, e, i);
                        if (ci < 0) continue;
                        for (int b = 0; b < nu; ++b)
                            for (int j = 0; j < pp; ++j) {
                                if (!structural(a, i, b, j)) continue;
                                const int32_t cj = colof(b, e, j);
                                if (cj >= 0) emit(ci, cj);
                            }
                    }
        };
        each_pair([&](int32_t ci, int32_t) { rc[ci + 1]++; });
        std::vector<int64_t> off(m + 1, 0);
        for (int64_t i = 0; i < m; ++i) off[i + 1] = off[i] + rc[i + 1];
        std::vector<int32_t> cols((size_t)off[m]);
        {
            std::vector<int64_t> fill(off.begin(), off.end() - 1);
            each_pair([&](int32_t ci, int32_t cj) { cols[(size_t)fill[ci]++] = cj; });
        }
        for (int64_t i = 0; i < m; ++i) {
            int32_t* lo = cols.data() + off[i];
            int32_t* hi = cols.data() + off[i + 1];
            std::sort(lo, hi);
            hi = std::unique(lo, hi);
            if (lo == hi) L.hHcol.push_back((int32_t)i);     // unknown untouched by any element: keep a diagonal slot
            else L.hHcol.insert(L.hHcol.end(), lo, hi);
            MGB_REQUIRE(L.hHcol.size() < (size_t)INT32_MAX, "Hessian pattern exceeds 32-bit indexing");
            L.hHptr[i + 1] = (int32_t)L.hHcol.size();
        }
    } else {
        // union over elements of (all columns of e) x (all columns of e)
        std::vector<int32_t> cnt(m + 1, 0);
        for (int64_t e = 0; e < NE; ++e)
            for (int32_t q = ecol_ptr[e * nu]; q < ecol_ptr[(e + 1) * nu]; ++q) cnt[ecols[q] + 1]++;
        for (int64_t j = 0; j < m; ++j) cnt[j + 1] += cnt[j];
        std::vector<int32_t> c2e(cnt[m]);
        {
            std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1);
            for (int64_t e = 0; e < NE; ++e)
                for (int32_t q = ecol_ptr[e * nu]; q < ecol_ptr[(e + 1) * nu]; ++q) c2e[fill[ecols[q]]++] = (int32_t)e;
        }
        std::vector<int32_t> rowbuf;
        for (int64_t i = 0; i < m; ++i) {
            rowbuf.clear();
            for (int32_t t = cnt[i]; t < cnt[i + 1]; ++t) {
                const int64_t e = c2e[t];
                rowbuf.insert(rowbuf.end(), ecols.begin() + ecol_ptr[e * nu], ecols.begin() + ecol_ptr[(e + 1) * nu]);
            }
            if (rowbuf.empty()) rowbuf.push_back((int32_t)i);   // unknown untouched by any element: keep a diagonal slot
            std::sort(rowbuf.begin(), rowbuf.end());
            rowbuf.erase(std::unique(rowbuf.begin(), rowbuf.end()), rowbuf.end());
            L.hHcol.insert(L.hHcol.end(), rowbuf.begin(), rowbuf.end());
            MGB_REQUIRE(L.hHcol.size() < (size_t)INT32_MAX, "Hessian pattern exceeds 32-bit indexing");
            L.hHptr[i + 1] = (int32_t)L.hHcol.size();
        }
    }
    if (!device_plan) {
        L.nnz = (int64_t)L.hHcol.size();
        L.Hptr.upload(L.hHptr, st);
        L.Hcol.upload(L.hHcol, st);
    }
    L.Hval.alloc((size_t)(L.nnz + L.m + 1));        // + the border column of the bordered factorization (mf_solver.hpp)
    L.selection = selection;
    const int NB = hel_blocks(nu);
    // 3. contribution lists: every structural nonzero of H gathers its summands from a slab in
    //    element order (no atomics).  Selection levels read the element-block slab written by
    //    the f2 kernel directly; general levels read the projected slab panel' * Hel * panel.
    auto find = [&](int32_t row, int32_t col) {
        const int32_t* lo = L.hHcol.data() + L.hHptr[row];
        const int32_t* hi = L.hHcol.data() + L.hHptr[row + 1];
        return (int32_t)(std::lower_bound(lo, hi, col) - L.hHcol.data());
    };
    int64_t sel_off[MGBHIP_MAX_NU * (MGBHIP_MAX_NU + 1) / 2];
    hel_layout(nu, NE, pp, diag_mask_sel, sel_off);
    auto for_each = [&](auto&& emit) {
        if (selection) {
            for (int64_t e = 0; e < NE; ++e)
                for (int a = 0; a < nu; ++a)
                    for (int i = 0; i < pp; ++i) {
                        const int32_t ci = colof(a, e, i);
                        if (ci < 0) continue;
                        for (int b = 0; b < nu; ++b)
                            for (int j = 0; j < pp; ++j) {
                                if (!structural(a, i, b, j)) continue;
                                const int32_t cj = colof(b, e, j);
                                if (cj < 0) continue;
                                int64_t src;   // slab index of Hel_ab[i, j] (upper block triangle stored)
                                const int blk = a <= b ? hel_block_index(a, b, nu) : hel_block_index(b, a, nu);
                                if ((diag_mask_sel >> blk) & 1) src = sel_off[blk] + e * pp + i;            // i == j here
                                else if (a <= b) src = sel_off[blk] + (e * pp + j) * (int64_t)pp + i;
                                else src = sel_off[blk] + (e * pp + i) * (int64_t)pp + j;
                                emit(find(ci, cj), src);
                            }
                    }
        } else {
            for (int64_t e = 0; e < NE; ++e) {
                const int32_t base = ecol_ptr[e * nu];
                const int32_t ct = ecol_ptr[(e + 1) * nu] - base;
                for (int32_t gj = 0; gj < ct; ++gj)
                    for (int32_t gi = 0; gi < ct; ++gi)
                        emit(find(ecols[base + gi], ecols[base + gj]), eoff[e] + gi + (int64_t)ct * gj);
            }
        }
    };
    if (!L.acc && !device_plan) {
        // one pass over the contributions: remember (position, source) pairs so that the binary
        // searches behind `find` run once, then bucket them by position
        std::vector<int32_t> ccount(L.nnz + 1, 0);
        std::vector<int32_t> ppos, psrc;
        for_each([&](int32_t pos, int64_t src) {
            MGB_REQUIRE(src < (int64_t)INT32_MAX, "slab exceeds 32-bit indexing");
            ccount[pos + 1]++;
            ppos.push_back(pos);
            psrc.push_back((int32_t)src);
        });
        int64_t total = 0;
        for (int64_t q = 0; q < L.nnz; ++q) {
            total += ccount[q + 1];
            MGB_REQUIRE(total < (int64_t)INT32_MAX, "contribution list exceeds 32-bit indexing");
            ccount[q + 1] += ccount[q];
        }
        L.long_lists = L.nnz > 0 && total / L.nnz > 48;
        std::vector<int32_t> fill(ccount.begin(), ccount.end() - 1);
        std::vector<int32_t> cidx((size_t)total);
        for (size_t t = 0; t < ppos.size(); ++t) cidx[fill[ppos[t]]++] = psrc[t];   // element order within a list
        L.cptr.upload(ccount, st);
        L.cidx.upload(cidx.data(), cidx.size(), st);
        if (cidx.empty()) L.cidx.alloc(1);
        MGB_HIP_CHECK(hipStreamSynchronize(st));
    }
    if (!L.acc && L.long_lists && L.nnz > 0) {
        // very long lists (average > 2048 contributions): split every list into <= 64 fixed chunks
        std::vector<int32_t> hc((size_t)L.nnz + 1);
        L.cptr.download(hc.data(), hc.size(), st);
        MGB_HIP_CHECK(hipStreamSynchronize(st));
        int64_t maxlen = 0;
        for (int64_t q = 0; q < L.nnz; ++q) maxlen = std::max<int64_t>(maxlen, hc[q + 1] - hc[q]);
        if (hc[L.nnz] / L.nnz > 2048) {
            L.gather_chunk = (int32_t)std::max<int64_t>(1024, (maxlen + 63) / 64);
            L.gather_nchunk = (int32_t)((maxlen + L.gather_chunk - 1) / L.gather_chunk);
            L.gather_part.alloc(2 * (size_t)L.nnz * (size_t)L.gather_nchunk);      // (sum, carried error) per chunk
        }
    }
    if (!selection && !L.acc && L.nnz > 0) {
        // positions of the upper triangle: `symmetric(H)` (src/newton.jl:253) and the factorization read nothing else, so the
        // Newton loop projects and gathers only those (half the slab stores, half the gather traffic)
        // ... in COLUMN-major order (column j, rows i <= j ascending): the summands of (i, j) and (i + 1, j) sit next to each
        // other in every element block that holds both (block entry ci + ct cj, columns sorted), so neighbouring waves of the
        // gather share their cache lines; in row-major order every 8-byte summand was a line of its own.  The order of the
        // POSITIONS changes, not the order inside a sum: same values.
        std::vector<int32_t> up;
        {
            std::vector<int32_t> cnt((size_t)m + 1, 0);
            for (int64_t i = 0; i < m; ++i)
                for (int32_t q = L.hHptr[i]; q < L.hHptr[i + 1]; ++q)
                    if (L.hHcol[q] >= i) cnt[(size_t)L.hHcol[q] + 1]++;
            for (int64_t j = 0; j < m; ++j) cnt[(size_t)j + 1] += cnt[(size_t)j];
            up.resize((size_t)cnt[(size_t)m]);
            static const bool row_major = [] { const char* e = getenv("MGBHIP_UPPER_ROW_MAJOR"); return e && e[0] == '1'; }();
            size_t w = 0;
            for (int64_t i = 0; i < m; ++i)
                for (int32_t q = L.hHptr[i]; q < L.hHptr[i + 1]; ++q)
                    if (L.hHcol[q] >= i) {
                        if (row_major) up[w++] = q;
                        else up[(size_t)cnt[(size_t)L.hHcol[q]]++] = q;
                    }
        }
        L.nup = (int64_t)up.size();
        L.upq.upload(up, st);
        MGB_HIP_CHECK(hipStreamSynchronize(st));
    }
    if (!selection) {
        // dense R panels per (element, state): p x c, column-major
        std::vector<double> panels((size_t)pp * ecols.size(), 0.0);
        for (int64_t e = 0; e < NE; ++e)
            for (int a = 0; a < nu; ++a) {
                const int32_t o = ecol_ptr[e * nu + a], c = ecol_ptr[e * nu + a + 1] - o;
                for (int r = 0; r < pp; ++r) {
                    const int64_t row = (int64_t)a * nn + e * pp + r;
                    for (int32_t q = L.hRptr[row]; q < L.hRptr[row + 1]; ++q) {
                        const int32_t* lo = ecols.data() + o;
                        const int32_t ia = (int32_t)(std::lower_bound(lo, lo + c, L.hRcol[q]) - lo);
                        panels[(size_t)pp * o + r + (size_t)pp * ia] += L.hRval[q];
                    }
                }
            }
        L.cmax = 1;
        for (size_t q = 0; q + 1 < ecol_ptr.size(); ++q) L.cmax = std::max(L.cmax, ecol_ptr[q + 1] - ecol_ptr[q]);
        std::vector<int32_t> eoff32(eoff.begin(), eoff.end());
        if (eoff32.empty()) eoff32.push_back(0);
        L.eoff.upload(eoff32, st);
        L.ecol_ptr.upload(ecol_ptr, st);
        L.ecols.upload(ecols.data(), ecols.size(), st);
        L.panels.upload(panels.data(), panels.size(), st);
        if (ecols.empty()) { L.ecols.alloc(1); L.panels.alloc(1); }
        L.slab.alloc((size_t)std::max<int64_t>(L.slab_doubles, 1));
        static const bool no_sorted = [] { const char* e = getenv("MGBHIP_NO_SORTED_SLAB"); return e && e[0] == '1'; }();
        if (!L.acc && L.nnz > 0 && L.slab_doubles > 0 && !no_sorted) {
            // Every entry of an element's projected block contributes to exactly one Hessian entry: the lists are a
            // permutation of the slab.  Store the slab in LIST order (the projection scatters through spos, 4 B per entry)
            // and a contribution list is a contiguous run: the gather streams it instead of fetching one cache line per
            // 8-byte summand (fem3d L = 6 phase I, m = 1 349: 23 M summands per assembly).
            int32_t total = 0;
            MGB_HIP_CHECK(hipMemcpyAsync(&total, L.cptr.p + L.nnz, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            MGB_HIP_CHECK(hipStreamSynchronize(st));
            if ((int64_t)total == L.slab_doubles) {
                L.spos.alloc((size_t)L.slab_doubles);
                launch_invert_lists(L.cidx.p, L.slab_doubles, L.spos.p, st);
                L.sorted_slab = true;
            }
        }
        if (L.acc) {
            L.acc_copies.alloc((size_t)L.acc_waves * (size_t)(m * (m + 1) / 2));
        }
        MGB_HIP_CHECK(hipStreamSynchronize(st));
    }
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    L.planned = true;
}

// ---------------------------------------------------------------------------------------------
// device-resident primitives
// ---------------------------------------------------------------------------------------------

ElemParams mgbhip_problem::base_params(int level, const double* d_s, const double* d_zz, const double* d_cc) const {
    ElemParams E;
    std::memset(&E, 0, sizeof(E));
    E.p = p; E.nu = nu; E.nD = nD; E.nstage = nstage; E.N = N; E.n = n;
    E.ymask = 0;
    for (int k = 0; k < cone.npieces; ++k)
        for (int c = 0; c < cone.pc[k].ni; ++c) E.ymask |= 1 << cone.pc[k].idx[c];
    if (cone.feasibility)
        for (int k = cone.NC - 1; k < nD; ++k) E.ymask |= 1 << k;
    for (size_t o = 0; o < store->ops.size() && o < MGBHIP_MAX_OPS; ++o) E.ops[o] = store->ops[o].p;
    for (int o = 0; o < nstage; ++o) E.stage_ptr[o] = stage_ptr[o];
    for (int k = 0; k < MGBHIP_MAX_ND; ++k) { E.D_state[k] = D_state[k]; E.D_op[k] = D_op[k]; E.D_stage[k] = D_stage[k]; }
    E.w = store->w.p;
    E.c = d_cc;
    // z0 + R*s is formed once per evaluation point by a row-parallel kernel (cached on the
    // (level, s, z) triple by the callers below); the element kernels read the result
    E.z0 = d_zz;
    if (level >= 0 && d_s != nullptr) {
        const Level& L = levels[level];
        if (zf_stamp == zstamp && zf_level == level && zf_s == d_s && zf_z == d_zz) {
            // same evaluation point as the previous call: d_zfull is still valid
        } else if (L.R_unit && !dense) {
            // (the dense spectral path, launch_dense_eval, reads E.z0 only: it never takes this branch)
            // selection level: the element kernel gathers s itself and leaves z0 + R s in d_zfull for the next evaluation
            // at this point (the Hessian at an accepted line-search trial reads it back instead of chaining two gathers)
            E.zsel = L.Rsel.p;
            E.zs = d_s;
            if (trial_x) { E.zs = trial_x; E.zx = trial_dir; E.zalpha = trial_alpha; }      // d_s (the trial vector) is written behind this launch
            E.zout = d_zfull.p;
            zf_stamp = zstamp; zf_level = level; zf_s = d_s; zf_z = d_zz;
            goto prolonged;
        } else if (L.R_long) {      // dense prolongation rows: wave per row
            MGB_HIP_CHECK(hipMemcpyAsync(d_zfull.p, d_zz, sizeof(double) * (size_t)L.rows, hipMemcpyDeviceToDevice, stream()));
            launch_csr_matvec(L.rows, L.Rptr.p, L.Rcol.p, L.Rval.p, d_s, d_zfull.p, true, true, stream());
        } else {
            launch_prolong(L.rows, L.Rptr.p, L.Rcol.p, L.Rval.p, d_s, d_zz, d_zfull.p, stream());
        }
        zf_stamp = zstamp; zf_level = level; zf_s = d_s; zf_z = d_zz;
        E.z0 = d_zfull.p;
    }
prolonged:
    E.bw = has_bw ? bw.p : nullptr;
    E.invn = 1.0 / (double)n;
    E.cone = cone;
    E.out_partial = d_partials.p;
    E.out_ret = d_ret.p;
    E.out_hel = d_hel.p;
    E.out_F = d_nodeF.p;
    E.out_Dz = nullptr;
    E.dn_Dz = d_dnDz.p;
    E.dn_Y = d_dnY.p;
    return E;
}

void mgbhip_problem::eval_f0_launch(int level, const double* d_s, const double* d_zz, const double* d_cc) {
    hipStream_t st = stream();
    StageScope sc(ctx->timers, level + 1 == (int)levels.size() ? "f0" : "f0_coarse");
    ElemParams E = base_params(level, d_s, d_zz, d_cc);
    launch_elem(E, MODE_F0, st);
    launch_reduce_partials(d_partials.p, elem_grid(p, N), d_scal.p, st);
    cnt.f0++;
}

double mgbhip_problem::eval_f0(int level, const double* d_s, const double* d_zz, const double* d_cc) {
    hipStream_t st = stream();
    eval_f0_launch(level, d_s, d_zz, d_cc);
    read_scalars(0, 1);
    if (sharded()) allreduce_host(pin.d, 1, 0);
    return pin.d[0];
}

void mgbhip_problem::allreduce_host(double* h, int64_t count, int op) {
    MGB_REQUIRE(coll_fn != nullptr, "no collective installed");
    if (coll_fn(coll_user, h, count, op, 0) != 0) throw mgbhip::InvalidArgument("the caller's allreduce failed");
}

void mgbhip_problem::allreduce_device(double* d, int64_t count, int op) {
    MGB_REQUIRE(coll_fn != nullptr, "no collective installed");
    hipStream_t st = stream();
    if (coll_device) {
        MGB_HIP_CHECK(hipStreamSynchronize(st));            // the buffer is final before the callee's stream touches it
        if (coll_fn(coll_user, d, count, op, 1) != 0) throw mgbhip::InvalidArgument("the caller's allreduce failed");
        return;
    }
    coll_host.resize((size_t)count);
    MGB_HIP_CHECK(hipMemcpyAsync(coll_host.data(), d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, st));
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    allreduce_host(coll_host.data(), count, op);
    MGB_HIP_CHECK(hipMemcpyAsync(d, coll_host.data(), sizeof(double) * (size_t)count, hipMemcpyHostToDevice, st));
    MGB_HIP_CHECK(hipStreamSynchronize(st));               // coll_host may be reused by the next call
}

void mgbhip_problem::reduce_interface(int level, double* d_g, double* d_part) {
    Level& L = levels[level];
    if (!L.sharded) return;
    hipStream_t st = stream();
    if (d_part) MGB_HIP_CHECK(hipMemcpyAsync(d_part, d_g, sizeof(double) * (size_t)L.m, hipMemcpyDeviceToDevice, st));
    const int64_t ni = (int64_t)L.h_iface.size();
    if (ni == 0) return;
    L.iface_buf.ensure((size_t)ni);
    launch_index_gather(d_g, L.d_iface.p, ni, L.iface_buf.p, st);
    allreduce_device(L.iface_buf.p, ni, 0);
    launch_index_scatter(L.iface_buf.p, L.d_iface.p, ni, d_g, st);
}

void mgbhip_problem::eval_f1(int level, const double* d_s, const double* d_zz, const double* d_cc, double* d_gout, double* d_part) {
    hipStream_t st = stream();
    const Level& L = levels[level];
    {
        StageScope sc(ctx->timers, level + 1 == (int)levels.size() ? "f1" : "f1_coarse");
        ElemParams E = base_params(level, d_s, d_zz, d_cc);
        launch_elem(E, MODE_F1, st);
    }
    {
        StageScope sc(ctx->timers, "restrict");
        if (L.T_chunks > 0)
            launch_csr_matvec_chunked(L.m, L.Tptr.p, L.Tcol.p, L.Tval.p, d_ret.p, d_gout, d_tchunk.p, L.T_chunks, st);
        else
            launch_csr_matvec(L.m, L.Tptr.p, L.Tcol.p, L.Tval.p, d_ret.p, d_gout, false, L.T_long, st);
    }
    if (L.sharded) reduce_interface(level, d_gout, d_part);
    cnt.f1++;
}

void mgbhip_problem::eval_f01_launch(int level, const double* d_s, const double* d_zz, const double* d_cc, double* d_gout,
                                     double* d_part, bool defer_f0_sum) {
    if (dense) {                    // the dense (spectral) path keeps its separate GEMV pipelines
        eval_f0_launch(level, d_s, d_zz, d_cc);
        eval_f1(level, d_s, d_zz, d_cc, d_gout, d_part);
        return;
    }
    hipStream_t st = stream();
    const Level& L = levels[level];
    {
        StageScope sc(ctx->timers, level + 1 == (int)levels.size() ? "f01" : "f01_coarse");
        ElemParams E = base_params(level, d_s, d_zz, d_cc);
        launch_elem(E, MODE_F01, st);
        if (!defer_f0_sum) launch_reduce_partials(d_partials.p, elem_grid(p, N), d_scal.p, st);
    }
    {
        StageScope sc(ctx->timers, "restrict");
        trial_fuse.done = false;
        if (trial_fuse.want && L.T_chunks == 0 && !L.T_long && !sharded()) {
            // line-search trial: restriction, |g|^2 partial sums and the step kernel's work in one launch (kernels.hip)
            launch_restrict_trial(L.m, L.Tptr.p, L.Tcol.p, L.Tval.p, d_ret.p, d_gout, d_scratch.p, trial_fuse.x, trial_fuse.n, trial_fuse.s,
                                  trial_fuse.xn, trial_fuse.moved, trial_fuse.stamp, st);
            trial_fuse.done = true;
        } else if (L.T_chunks > 0)
            launch_csr_matvec_chunked(L.m, L.Tptr.p, L.Tcol.p, L.Tval.p, d_ret.p, d_gout, d_tchunk.p, L.T_chunks, st);
        else
            launch_csr_matvec(L.m, L.Tptr.p, L.Tcol.p, L.Tval.p, d_ret.p, d_gout, false, L.T_long, st);
    }
    if (L.sharded) reduce_interface(level, d_gout, d_part);
    cnt.f0++;
    cnt.f1++;
}

bool mgbhip_problem::try_enable_condensed(int level) {
    Level& L = levels[level];
    static const bool off = [] { const char* e = getenv("MGBHIP_NO_CONDENSE"); return e && e[0] == '1'; }();
    if (off || dense || !L.selection || !L.direct || nu != 2 || p != 7 || L.hRptr.empty()) return false;
    if (!L.solver.analyzed || !L.solver.has_direct_map()) return false;
    // R of a selection level: row (state, node) has at most one entry, equal to 1.  Slack unknowns must be numbered
    // slack0 + node (R_fine = blockdiag(R_dirichlet, I), src/multigrid.jl:474-512).
    const int64_t slack0 = L.m - n;
    if (slack0 < 0 || L.rows != 2 * n) return false;
    std::vector<int32_t> ucol((size_t)n, -1);
    for (int64_t node = 0; node < n; ++node) {
        const int32_t b = L.hRptr[node], e = L.hRptr[node + 1];
        if (e - b > 1) return false;
        if (e - b == 1) ucol[(size_t)node] = L.hRcol[b];
        const int64_t rs = n + node;
        if (L.hRptr[rs + 1] - L.hRptr[rs] != 1 || (int64_t)L.hRcol[L.hRptr[rs]] != slack0 + node) return false;
    }
    const bool ok = L.solver.enable_condensed(N, p, ucol.data(), slack0, L.nnz, hel_cap + L.nshared, stream());
    if (const char* dbg = getenv("MGBHIP_DEBUG"); dbg && atoi(dbg) >= 2)
        fprintf(stderr, "[mgbhip] level %d: condensed leaves %s\n", level, ok ? "enabled" : "not applicable");
    return ok;
}

void mgbhip_problem::eval_f2(int level, const double* d_s, const double* d_zz, const double* d_cc, bool materialize,
                             const double* rhs) {
    ensure_plan(level);
    hipStream_t st = stream();
    Level& L = levels[level];
    int E_ymask = 0;
    if (L.condense && !materialize && rhs != nullptr && L.selection && L.direct && !dense) {
        // Newton loop on a level with condensed leaves: one kernel evaluates the element blocks and eliminates the
        // element-local unknowns; nothing else is assembled (the other fronts receive every contribution through the
        // leaves' update blocks)
        StageScope sc(ctx->timers, level + 1 == (int)levels.size() ? "f2" : "f2_coarse");
        ElemParams E = base_params(level, d_s, d_zz, d_cc);
        E.leaf_desc = L.solver.leaf_desc();
        E.leaf_arena = L.solver.arena();
        E.leaf_g = rhs;
        E.leaf_slack0 = L.m - n;
        E.leaf_status = L.solver.leaf_status();
        E.leaf_packed = L.solver.leaves_packed() ? 1 : 0;
        if (!L.solver.leaf_flag_zero()) MGB_HIP_CHECK(hipMemsetAsync(E.leaf_status, 0, sizeof(int32_t), st));
        L.solver.leaf_flag_used();
        if (launch_elem_f2_condense(E, st)) {
            L.have_H = true;
            L.H_in_slab = true;
            L.H_condensed = true;
            L.condensed_rhs = rhs;
            hel_level = level;
            L.factored = false;
            cnt.f2++;
            return;
        }
        zf_stamp = -1;              // nothing ran: base_params may have promised d_zfull to a kernel that was not launched
    }
    L.H_condensed = false;
    {
        // fine-level launches are timed apart: they are the ones the HBM roofline is quoted on
        StageScope sc(ctx->timers, level + 1 == (int)levels.size() ? "f2" : "f2_coarse");
        ElemParams E = base_params(level, d_s, d_zz, d_cc);
        E_ymask = E.ymask;
        E.diag_mask = (L.selection && !dense) ? diag_mask_sel : 0;
        hel_layout(nu, N, p, E.diag_mask, E.blk_off);
        launch_elem(E, MODE_F2, st);
    }
    {
        StageScope sc(ctx->timers, level + 1 == (int)levels.size() ? "assemble" : "assemble_coarse");
        if (dense) {
            // rows of y that enter the barrier form a contiguous range in every supported D
            // layout; rows outside it have zero weights and are skipped in the product
            int klo = nD, khi = -1;
            for (int k = 0; k < nD; ++k)
                if ((E_ymask >> k) & 1) { klo = std::min(klo, k); khi = std::max(khi, k); }
            const int64_t ld = (int64_t)nD * n;
            if (khi < klo) {
                MGB_HIP_CHECK(hipMemsetAsync(L.Hval.p, 0, sizeof(double) * (size_t)L.nnz, st));
            } else {
                launch_dense_weight(nD, klo, khi, n, L.m, ld, L.denseDR.p, d_dnY.p, L.denseW.p, st);
                launch_dense_gemm_tn((int)L.m, (int)L.m, (int)((khi - klo + 1) * n), L.denseDR.p + (int64_t)klo * n, ld,
                                     nullptr, L.denseW.p + (int64_t)klo * n, ld, L.Hval.p, L.m, false, true, st);
            }
        } else if (L.selection && L.direct && !materialize) {
            launch_gather_shared(L.nshared, L.sh_q.p, L.cptr.p, L.cidx.p, d_hel.p, d_hel.p + hel_cap, st);
        } else if (L.selection) {
            launch_gather_assemble(L.nnz, L.cptr.p, L.cidx.p, d_hel.p, L.Hval.p, L.long_lists, st, L.gather_chunk, L.gather_nchunk, L.gather_part.p);
        } else {
            PanelParams PP;
            PP.p = p; PP.nu = nu; PP.N = N;
            PP.ecol_ptr = L.ecol_ptr.p; PP.panels = L.panels.p; PP.eoff = L.eoff.p;
            PP.hel = d_hel.p; PP.slab = L.slab.p; PP.cmax = L.cmax; PP.spos = nullptr;
            // Newton loop (materialize = false with a right-hand side): only the upper triangle of H is formed
            static const bool full_h = [] { const char* e = getenv("MGBHIP_FULL_COARSE_H"); return e && e[0] == '1'; }();
            const bool upper = !materialize && rhs != nullptr && L.nup > 0 && !L.acc && !full_h;
            PP.upper_only = upper ? 1 : 0;
            if (L.acc) {
                launch_panel_accumulate(PP, L.ecols.p, (int32_t)L.m, L.acc_waves, L.acc_split, L.acc_chunk, L.acc_ctmax,
                                        L.acc_copies.p, L.Hval.p, st);
            } else {
                PP.spos = L.sorted_slab ? L.spos.p : nullptr;
                // the two products of the projection on the matrix cores; the loop kernels remain for elements whose
                // staging does not fit (staged variant while four workgroups still fit a CU: narrow supports)
                if (launch_panel_project_mfma(PP, st)) {
                } else if (panel_accumulate_lds(p, nu, L.acc_ctmax) <= 40 * 1024) launch_panel_project_staged(PP, L.acc_ctmax, st);
                else launch_panel_project(PP, st);
                launch_gather_assemble(L.nnz, L.cptr.p, L.sorted_slab ? nullptr : L.cidx.p, L.slab.p, L.Hval.p, L.long_lists, st, L.gather_chunk,
                                       L.gather_nchunk, L.gather_part.p, upper ? L.upq.p : nullptr, upper ? L.nup : 0);
            }
        }
    }
    L.have_H = true;
    L.H_in_slab = L.selection && L.direct && !materialize && !dense;
    hel_level = L.H_in_slab ? level : -1;         // any f2 overwrites the slab
    L.factored = false;
    cnt.f2++;
}

void mgbhip_problem::wait_results(double seq) {
    static const bool poll = [] { const char* e = getenv("MGBHIP_NO_POLL"); return !(e && e[0] == '1'); }();
    hipStream_t st = stream();
    if (poll && seq != 0.0) {
        volatile const double* stamp = pin.d + 15;
        const auto t0 = std::chrono::steady_clock::now();
        for (int spins = 0;; ++spins) {
            if (__atomic_load_n(reinterpret_cast<const volatile uint64_t*>(stamp), __ATOMIC_ACQUIRE) ==
                *reinterpret_cast<const uint64_t*>(&seq))
                return;
            if ((spins & 1023) == 1023 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 1e-3)
                break;                                   // a long kernel or an error: let the runtime wait (and report)
        }
    }
    MGB_HIP_CHECK(hipStreamSynchronize(st));
}

void mgbhip_problem::read_scalars(int lo, int n) {
    const double seq = next_seq();
    launch_publish(d_scal.p + lo, n, pin.dev, lo, seq, stream());
    wait_results(seq);
}

bool mgbhip_problem::can_fuse_step(int level) const {
    static const bool off = [] { const char* e = getenv("MGBHIP_NO_FUSED_STEP"); return e && e[0] == '1'; }();
    return !off && !dense && !sharded() && levels[level].R_unit;
}

void mgbhip_problem::ensure_analysis(int level) {
    Level& L = levels[level];
    if (L.solver.analyzed) return;
    hipStream_t st = stream();
    const auto t0 = std::chrono::steady_clock::now();
    // ordering hint: the centroid of every level-J basis function, sum_i |R_ij| x_i / sum_i |R_ij|
    std::vector<double> cen;
    if (!hx.empty() && xdim > 0 && !L.hRptr.empty()) {
        cen.assign((size_t)L.m * xdim, 0.0);
        std::vector<double> wsum((size_t)L.m, 0.0);
        for (int64_t r = 0; r < L.rows; ++r) {
            const int64_t node = r % n;
            for (int32_t q = L.hRptr[r]; q < L.hRptr[r + 1]; ++q) {
                const double a = std::fabs(L.hRval[q]);
                const int32_t j = L.hRcol[q];
                wsum[j] += a;
                for (int d = 0; d < xdim; ++d) cen[(size_t)j * xdim + d] += a * hx[(size_t)d * n + node];
            }
        }
        for (int64_t j = 0; j < L.m; ++j)
            for (int d = 0; d < xdim; ++d) cen[(size_t)j * xdim + d] = wsum[j] > 0 ? cen[(size_t)j * xdim + d] / wsum[j] : 0.0;
    }
    // candidates for condensed leaves (try_enable_condensed) keep their per-element leaf fronts unmerged
    const bool leaves = L.selection && L.direct && !dense && nu == 2 && p == 7;
    if (L.sharded) {
        mgbhip_problem* self = this;
        L.solver.iface_reduce = [self](double* d, int64_t cnt2) { self->allreduce_device(d, cnt2, 0); };
    }
    L.solver.analyze(L.m, L.hHptr.data(), L.hHcol.data(), st, cen.empty() ? nullptr : cen.data(), xdim, leaves,
                     L.sharded ? L.h_iface.data() : nullptr, L.sharded ? (int64_t)L.h_iface.size() : 0);
    if (const char* dbg = getenv("MGBHIP_DEBUG"); dbg && atoi(dbg) >= 2)
        fprintf(stderr, "[mgbhip] symbolic analysis level %d (m=%lld, nnz=%lld): %.2f s\n", level, (long long)L.m,
                (long long)L.nnz, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
}

// Everything the first Newton iteration of every level would build lazily -- assembly plan, symbolic analysis, direct value map,
// condensed-leaf descriptors -- built now, the levels side by side on host threads (they are independent; the finest level's
// analysis is the long pole and starts first).  mgb_core calls this once per problem image: time to first solution.
void mgbhip_problem::prepare_all() {
    if (prepared) return;
    prepared = true;
    static const bool off = [] { const char* e = getenv("MGBHIP_LAZY_PLANS"); return e && e[0] == '1'; }();
    if (off || dense || sharded() || levels.size() < 2) return;     // sharded levels order their collectives: stay lazy
    const int Ln = (int)levels.size();
    const auto t_prep0 = std::chrono::steady_clock::now();
    std::vector<std::exception_ptr> errors((size_t)Ln);
    std::atomic<int> next{Ln - 1};
    auto worker = [&] {
        (void)hipSetDevice(ctx->device);
        for (int l = next.fetch_sub(1); l >= 0; l = next.fetch_sub(1)) {
            try {
                Level& L = levels[l];
                if (L.sharded) continue;
                ensure_plan(l);
                if (L.m < 1) continue;
                ensure_analysis(l);
                ensure_direct(l);
            } catch (...) {
                errors[(size_t)l] = std::current_exception();
            }
        }
    };
    // three workers: the finest level's chain (plan -> analysis on up to 16 host threads of its own -> direct map) is the
    // critical path, 0.5-0.6 s at L = 9; the other levels' 0.8 s of host work fit beside it on two workers without taking its
    // cores (first solve at L = 9 with 2 / 3 / 4 / 6 workers: 1.57 / 1.36 / 1.48 / 1.44 s)
    const int nthreads = std::min(Ln, 3);
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    if (const char* dbg = getenv("MGBHIP_DEBUG"); dbg && atoi(dbg) >= 2)
        fprintf(stderr, "[mgbhip] prepare_all: %d levels in %.3f s\n", Ln,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_prep0).count());
    // A level that could not be prepared (its plan exceeds an index range, the device is out of memory) is left to the lazy
    // path: the solve may never visit it (levels 8 and 9 of the L = 9 ladder are not), and if it does the error is raised there.
    for (size_t l = 0; l < errors.size(); ++l) {
        if (!errors[l]) continue;
        (void)hipGetLastError();
        if (const char* dbg = getenv("MGBHIP_DEBUG"); dbg && atoi(dbg) >= 1) {
            try { std::rethrow_exception(errors[l]); }
            catch (const std::exception& ex) { fprintf(stderr, "[mgbhip] prepare: level %zu left to the lazy path (%s)\n", l, ex.what()); }
            catch (...) { fprintf(stderr, "[mgbhip] prepare: level %zu left to the lazy path\n", l); }
        }
    }
}

// Direct value map (the factorization reads the element-block slab, H is never materialised) and the condensed leaves of a
// selection level: needs the plan and the analysis.
void mgbhip_problem::ensure_direct(int level) {
    Level& L = levels[level];
    if (!(L.selection && L.direct) || dense || !L.solver.analyzed) return;
    hipStream_t st = stream();
    const bool dbg2 = [] { const char* e = getenv("MGBHIP_DEBUG"); return e && atoi(e) >= 2; }();
    if (!L.solver.has_direct_map()) {
        const auto t0 = std::chrono::steady_clock::now();
        L.solver.set_direct_map(L.h_vmap.data(), L.nnz, hel_cap + L.nshared, st);
        std::vector<int32_t>().swap(L.h_vmap);        // 4 B per nonzero: not needed again
        if (dbg2) fprintf(stderr, "[mgbhip] direct value map level %d: %.2f s\n", level,
                          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    if (!L.condense_tried) {          // from the next f2 on the element kernel writes the leaf fronts itself
        const auto t0 = std::chrono::steady_clock::now();
        L.condense_tried = true;
        L.condense = try_enable_condensed(level);
        if (dbg2) fprintf(stderr, "[mgbhip] condensed-leaf set-up level %d: %.2f s\n", level,
                          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
}

void mgbhip_problem::factor(int level, const double* rhs) {
    Level& L = levels[level];
    MGB_REQUIRE(L.have_H, "solve requested before any Hessian was assembled at this level");
    hipStream_t st = stream();
    ensure_analysis(level);
    if (L.H_in_slab) {
        // H was not materialised (eval_f2 with materialize = false): the values are the slab + shared sums in d_hel
        MGB_REQUIRE(hel_level == level, "the element blocks of this level's Hessian were overwritten: evaluate f2 again");
        MGB_REQUIRE(rhs != nullptr, "a Hessian kept in the slab is factored together with its right-hand side");
        ensure_direct(level);
        MGB_REQUIRE(!L.H_condensed || rhs == L.condensed_rhs, "condensed leaves were formed for another right-hand side");
        double* tail = d_hel.p + hel_cap + L.nshared;
        launch_border_tail(rhs, tail, L.m, st);
        L.solver.factor(d_hel.p, st, &ctx->timers, true, L.H_condensed);
        L.border_state2 = 2;
        L.factored = true;
        cnt.factor++;
        return;
    }
    L.border_state2 = 0;
    if (rhs) {          // border column -g, corner -1: the factorization carries the forward substitution of H x = g
        launch_border_tail(rhs, L.Hval.p + L.nnz, L.m, st);
        L.border_state = 2;
    } else if (L.border_state != 1) {       // border column 0, corner 1: block diagonal, ordinary solves
        MGB_HIP_CHECK(hipMemsetAsync(L.Hval.p + L.nnz, 0, sizeof(double) * (size_t)L.m, st));
        launch_fill(1.0, L.Hval.p + L.nnz + L.m, 1, st);
        L.border_state = 1;
    }
    L.solver.factor(L.Hval.p, st, &ctx->timers);
    L.factored = true;
    cnt.factor++;
}

bool mgbhip_problem::lu_fallback(int level, const double* d_g, double* d_xout) {
    Level& L = levels[level];
    static const bool off = [] { const char* e = getenv("MGBHIP_NO_LU_FALLBACK"); return e && e[0] == '1'; }();
    if (off || sharded() || !L.have_H || L.H_in_slab || L.m < 1 || L.m > DENSE_LU_MAX_M || L.Hval.n < (size_t)L.nnz || !L.Hptr.p || !L.Hcol.p)
        return false;
    hipStream_t st = stream();
    d_lu.ensure((size_t)L.m * (size_t)L.m);
    d_lustat.ensure(1);
    d_lustat.zero(st, 1);
    launch_dense_lu_solve((int)L.m, L.Hptr.p, L.Hcol.p, L.Hval.p, d_lu.p, d_g, d_xout, d_lustat.p, st);
    int32_t bad = 1;
    d_lustat.download(&bad, 1, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    if (bad == 0) ++lu_fallbacks;
    return bad == 0;
}

void mgbhip_problem::trisolve(int level, const double* d_g, double* d_xout) {
    Level& L = levels[level];
    MGB_REQUIRE(L.factored, "triangular solve before factorization");
    MGB_REQUIRE(L.border_state == 1 && L.border_state2 == 0, "triangular solve on factors that carry a Newton right-hand side");
    L.solver.solve(d_g, d_xout, stream(), &ctx->timers);
}

void mgbhip_problem::trisolve_carried(int level, double* d_xout_np1) {
    Level& L = levels[level];
    MGB_REQUIRE(L.factored && (L.border_state == 2 || L.border_state2 == 2), "carried solve without a factorization that carries the right-hand side");
    L.solver.solve_border(d_xout_np1, stream(), &ctx->timers);
}
