// api.cpp -- extern "C" surface of libmgbhip.so (include/mgbhip.h).  Every entry point
// converts C++ exceptions to a status code + mgbhip_last_error(); nothing here computes on
// the CPU: a missing / failing GPU is an error, never a fallback.
#include <chrono>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <limits>
#include <memory>
#include <string>

#include "problem.hpp"

using namespace mgbhip;

mgbhip_problem* problem_create(mgbhip_ctx* ctx, const mgbhip_problem_desc* d, mgbhip_problem* share);
int core_run(mgbhip_problem* P, double* z, const double* c, const mgbhip_options* opt, mgbhip_core_result* res);
int matched_t_run(mgbhip_problem* P, const double* z, const double* c, double t_default, double* t_out);

#ifdef MGB_STEP_PROBE
namespace mgbhip { void mf_debug_probe(long long* out64); }
#endif

static thread_local std::string g_last_error;

// Every entry point that takes a handle runs with the handle's device current on the calling
// thread (and restores the caller's device afterwards): a second context on another GPU, a call
// from another host thread or a torch.cuda.set_device between calls must not put buffers or
// launches on the wrong device for ctx->stream.
struct DeviceGuard {
    int prev = -1, dev = -1;
    explicit DeviceGuard(int d) : dev(d) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) MGB_HIP_CHECK(hipSetDevice(dev));
    }
    ~DeviceGuard() {
        if (dev >= 0 && prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};
static int dev_of(const mgbhip_ctx* c) { return c ? c->device : -1; }
static int dev_of(const mgbhip_problem* p) { return (p && p->ctx) ? p->ctx->device : -1; }
static int dev_of(const mgbhip_vec* v) { return (v && v->ctx) ? v->ctx->device : -1; }

#define MGB_API_BEGIN try {
#define MGB_API_BEGIN_ON(h) try { DeviceGuard _guard(dev_of(h));
#define MGB_API_END                                   \
    }                                                 \
    catch (const InvalidArgument& e) {                \
        g_last_error = e.what();                      \
        return MGBHIP_ERR_INVALID;                    \
    }                                                 \
    catch (const HipError& e) {                       \
        g_last_error = e.what();                      \
        return MGBHIP_ERR_HIP;                        \
    }                                                 \
    catch (const std::exception& e) {                 \
        g_last_error = e.what();                      \
        return MGBHIP_ERR_INVALID;                    \
    }

extern "C" {

const char* mgbhip_last_error(void) { return g_last_error.c_str(); }
const char* mgbhip_version(void) { return "mgbhip 0.1 (gfx950)"; }

int mgbhip_create(mgbhip_ctx** out, int device_id, void* hip_stream) {
    MGB_API_BEGIN
    MGB_REQUIRE(out != nullptr, "null output pointer");
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    const bool dbg2 = [] { const char* e = getenv("MGBHIP_DEBUG"); return e && atoi(e) >= 2; }();
    int count = 0;
    MGB_HIP_CHECK(hipGetDeviceCount(&count));
    MGB_REQUIRE(count > 0, "no HIP device visible: this library has no CPU fallback");
    MGB_REQUIRE(device_id >= 0 && device_id < count, "device id out of range");
    MGB_HIP_CHECK(hipSetDevice(device_id));
    if (dbg2) fprintf(stderr, "[mgbhip] create: device selected after %.3f s\n", since());
    mgbhip_ctx* c = new mgbhip_ctx();
    c->device = device_id;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        MGB_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    c->timers.stream = c->stream;
    if (dbg2) fprintf(stderr, "[mgbhip] create: stream ready after %.3f s\n", since());
    *out = c;
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_destroy(mgbhip_ctx* ctx) {
    MGB_API_BEGIN_ON(ctx)
    if (!ctx) return MGBHIP_OK;
    (void)hipStreamSynchronize(ctx->stream);
    ctx->timers.reset(false);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_problem_create(mgbhip_ctx* ctx, const mgbhip_problem_desc* desc, mgbhip_problem* share,
                          mgbhip_problem** out) {
    MGB_API_BEGIN_ON(ctx)
    MGB_REQUIRE(out != nullptr, "null output pointer");
    *out = problem_create(ctx, desc, share);
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_problem_destroy(mgbhip_problem* prob) {
    MGB_API_BEGIN_ON(prob)
    if (!prob) return MGBHIP_OK;
    (void)hipStreamSynchronize(prob->stream());
    delete prob;
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_problem_set_box(mgbhip_problem* prob, double b, double R) {
    MGB_API_BEGIN
    MGB_REQUIRE(prob && prob->cone.feasibility, "set_box on a problem without the phase-I wrapper");
    prob->cone.box_b = b;
    prob->cone.box_R = R;
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_problem_set_barrier_weights(mgbhip_problem* prob, const double* bw) {
    MGB_API_BEGIN_ON(prob)
    MGB_REQUIRE(prob, "null problem");
    if (bw) {
        prob->bw.upload(bw, (size_t)prob->n, prob->stream());
        MGB_HIP_CHECK(hipStreamSynchronize(prob->stream()));
        prob->has_bw = true;
    } else {
        prob->has_bw = false;
    }
    return MGBHIP_OK;
    MGB_API_END
}

int64_t mgbhip_level_size(const mgbhip_problem* prob, int32_t level) {
    if (!prob || level < 0 || level >= (int32_t)prob->levels.size()) return -1;
    return prob->levels[level].m;
}

static void check_level(mgbhip_problem* P, int32_t level) {
    MGB_REQUIRE(P != nullptr, "null problem");
    MGB_REQUIRE(level >= 0 && level < (int32_t)P->levels.size(), "level out of range");
}

static void stage_inputs(mgbhip_problem* P, int32_t level, const double* s, const double* c, const double* z0) {
    hipStream_t st = P->stream();
    P->d_x.upload(s, (size_t)P->levels[level].m, st);
    P->d_c.upload(c, (size_t)P->n * P->nD, st);
    P->d_z0.upload(z0, (size_t)P->nu * P->n, st);
    P->touch();
}

int mgbhip_f0(mgbhip_problem* P, int32_t level, const double* s, const double* c, const double* z0, double* value) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    MGB_REQUIRE(s && c && z0 && value, "null argument");
    stage_inputs(P, level, s, c, z0);
    *value = P->eval_f0(level, P->d_x.p, P->d_z0.p, P->d_c.p);
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_f1(mgbhip_problem* P, int32_t level, const double* s, const double* c, const double* z0, double* grad) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    MGB_REQUIRE(s && c && z0 && grad, "null argument");
    stage_inputs(P, level, s, c, z0);
    P->eval_f1(level, P->d_x.p, P->d_z0.p, P->d_c.p, P->d_g.p);
    P->d_g.download(grad, (size_t)P->levels[level].m, P->stream());
    MGB_HIP_CHECK(hipStreamSynchronize(P->stream()));
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_f2(mgbhip_problem* P, int32_t level, const double* s, const double* c, const double* z0, double* values) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    MGB_REQUIRE(s && c && z0, "null argument");
    stage_inputs(P, level, s, c, z0);
    P->eval_f2(level, P->d_x.p, P->d_z0.p, P->d_c.p);
    if (values) P->levels[level].Hval.download(values, (size_t)P->levels[level].nnz, P->stream());
    MGB_HIP_CHECK(hipStreamSynchronize(P->stream()));
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_hessian_pattern(mgbhip_problem* P, int32_t level, int64_t* nnz, const int32_t** rowptr,
                           const int32_t** colidx) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    P->ensure_plan(level);
    if (nnz) *nnz = P->levels[level].nnz;
    if (rowptr) *rowptr = P->levels[level].hHptr.data();
    if (colidx) *colidx = P->levels[level].hHcol.data();
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_solve(mgbhip_problem* P, int32_t level, const double* g, double* x) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    MGB_REQUIRE(g && x, "null argument");
    hipStream_t st = P->stream();
    const size_t m = (size_t)P->levels[level].m;
    P->d_g.upload(g, m, st);
    P->factor(level);
    P->trisolve(level, P->d_g.p, P->d_nv.p);
    int status = P->levels[level].solver.status(st);
    if (status != MGBHIP_OK && P->lu_fallback(level, P->d_g.p, P->d_nv.p)) status = MGBHIP_OK;   // LDL' failed: pivoted LU (src/utils.jl:145)
    P->d_nv.download(x, m, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    if (status != MGBHIP_OK) g_last_error = "Cholesky met a non-positive pivot";
    return status;
    MGB_API_END
}

int mgbhip_set_hessian(mgbhip_problem* P, int32_t level, const double* values) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    MGB_REQUIRE(values, "null argument");
    P->ensure_plan(level);
    mgbhip::Level& L = P->levels[level];
    MGB_REQUIRE(L.Hval.n >= (size_t)L.nnz, "this level keeps no CSR value array");
    L.Hval.upload(values, (size_t)L.nnz, P->stream());
    MGB_HIP_CHECK(hipStreamSynchronize(P->stream()));
    L.have_H = true;
    L.H_in_slab = false;
    L.factored = false;
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_solve_newton(mgbhip_problem* P, int32_t level, const double* g, double* x, double* lambda2) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    MGB_REQUIRE(g && x, "null argument");
    hipStream_t st = P->stream();
    const size_t m = (size_t)P->levels[level].m;
    P->d_g.upload(g, m, st);
    P->factor(level, P->d_g.p);                    // [H -g; -g' -1]: the forward substitution rides along
    P->trisolve_carried(level, P->d_nv.p);         // one backward sweep from x_n = 1
    int status = P->levels[level].solver.status(st);
    if (status != MGBHIP_OK && P->lu_fallback(level, P->d_g.p, P->d_nv.p)) status = MGBHIP_OK;   // LDL' failed: pivoted LU (src/utils.jl:145)
    P->d_nv.download(x, m, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    if (lambda2) {
        double acc = 0.0;
        for (size_t i = 0; i < m; ++i) acc += g[i] * x[i];
        *lambda2 = acc;
    }
    if (status != MGBHIP_OK) g_last_error = "Cholesky met a non-positive pivot";
    return status;
    MGB_API_END
}

int mgbhip_problem_set_sharding(mgbhip_problem* P, int32_t level, int64_t n_iface, const int32_t* iface_cols, const double* own_mask) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    mgbhip::Level& L = P->levels[level];
    MGB_REQUIRE(!L.solver.analyzed, "sharding must be set before the first solve of the level");
    MGB_REQUIRE(own_mask != nullptr && n_iface >= 0 && n_iface <= L.m && (n_iface == 0 || iface_cols), "bad sharding arguments");
    L.h_iface.assign(iface_cols, iface_cols + n_iface);
    for (int64_t i = 0; i < n_iface; ++i) {
        MGB_REQUIRE(L.h_iface[i] >= 0 && L.h_iface[i] < L.m, "interface column out of range");
        MGB_REQUIRE(i == 0 || L.h_iface[i] > L.h_iface[i - 1], "interface columns must be strictly increasing");
    }
    L.d_iface.upload(L.h_iface, P->stream());
    L.own.upload(own_mask, (size_t)L.m, P->stream());
    MGB_HIP_CHECK(hipStreamSynchronize(P->stream()));
    L.sharded = true;
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_problem_set_collective(mgbhip_problem* P, mgbhip_allreduce_fn fn, void* user, int32_t accepts_device_ptr) {
    MGB_API_BEGIN_ON(P)
    P->coll_fn = fn;
    P->coll_user = user;
    P->coll_device = accepts_device_ptr != 0;
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_newton_direction(mgbhip_problem* P, int32_t level, const double* s, const double* c, const double* z0, double* x,
                            double* lambda2, int32_t* condensed) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    MGB_REQUIRE(s && c && z0 && x, "null argument");
    hipStream_t st = P->stream();
    const size_t m = (size_t)P->levels[level].m;
    stage_inputs(P, level, s, c, z0);
    P->eval_f1(level, P->d_x.p, P->d_z0.p, P->d_c.p, P->d_g.p);
    P->eval_f2(level, P->d_x.p, P->d_z0.p, P->d_c.p, false, P->d_g.p);      // exactly the Newton loop's sequence
    if (condensed) *condensed = P->levels[level].H_condensed ? 1 : 0;
    P->factor(level, P->d_g.p);
    P->trisolve_carried(level, P->d_nv.p);
    const int status = P->levels[level].solver.status(st);
    P->d_nv.download(x, m, st);
    if (lambda2) {
        std::vector<double> g(m);
        P->d_g.download(g.data(), m, st);
        MGB_HIP_CHECK(hipStreamSynchronize(st));
        double acc = 0.0;
        for (size_t i = 0; i < m; ++i) acc += g[i] * x[i];
        *lambda2 = acc;
    }
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    if (status != MGBHIP_OK) g_last_error = "Cholesky met a non-positive pivot";
    return status;
    MGB_API_END
}


// ---- device-resident vectors --------------------------------------------------------------------
static double* vec_scratch(mgbhip_ctx* c, int64_t len) {
    c->vscratch.ensure((size_t)reduce_scratch_doubles(len));
    c->vscal.ensure(8);
    return c->vscratch.p;
}
static void same_ctx(const mgbhip_vec* a, const mgbhip_vec* b) {
    MGB_REQUIRE(a && b, "null vector");
    MGB_REQUIRE(a->ctx == b->ctx, "vectors belong to different contexts");
}

int mgbhip_vec_alloc(mgbhip_ctx* ctx, int64_t len, mgbhip_vec** out) {
    MGB_API_BEGIN_ON(ctx)
    MGB_REQUIRE(ctx && out && len >= 0, "bad argument");
    std::unique_ptr<mgbhip_vec> v(new mgbhip_vec());
    v->ctx = ctx;
    v->len = len;
    v->buf.alloc((size_t)std::max<int64_t>(len, 1));
    v->buf.zero(ctx->stream);
    *out = v.release();
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_vec_free(mgbhip_vec* v) {
    MGB_API_BEGIN_ON(v)
    if (!v) return MGBHIP_OK;
    (void)hipStreamSynchronize(v->ctx->stream);
    delete v;
    return MGBHIP_OK;
    MGB_API_END
}

int64_t mgbhip_vec_len(const mgbhip_vec* v) { return v ? v->len : -1; }

int mgbhip_vec_upload(mgbhip_vec* v, const double* host, int64_t len) {
    MGB_API_BEGIN_ON(v)
    MGB_REQUIRE(v && host && len == v->len, "upload: length mismatch");
    v->buf.upload(host, (size_t)len, v->ctx->stream);
    MGB_HIP_CHECK(hipStreamSynchronize(v->ctx->stream));     // the host buffer may go away
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_vec_download(const mgbhip_vec* v, double* host, int64_t len) {
    MGB_API_BEGIN_ON(v)
    MGB_REQUIRE(v && host && len == v->len, "download: length mismatch");
    v->buf.download(host, (size_t)len, v->ctx->stream);
    MGB_HIP_CHECK(hipStreamSynchronize(v->ctx->stream));
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_vec_fill(mgbhip_vec* v, double value) {
    MGB_API_BEGIN_ON(v)
    MGB_REQUIRE(v, "null vector");
    launch_fill(value, v->buf.p, v->len, v->ctx->stream);
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_vec_copy(mgbhip_vec* dst, const mgbhip_vec* src) {
    MGB_API_BEGIN_ON(dst)
    same_ctx(dst, src);
    MGB_REQUIRE(dst->len == src->len, "copy: length mismatch");
    if (dst->len) MGB_HIP_CHECK(hipMemcpyAsync(dst->buf.p, src->buf.p, (size_t)dst->len * sizeof(double),
                                               hipMemcpyDeviceToDevice, dst->ctx->stream));
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_vec_axpy(double alpha, const mgbhip_vec* x, mgbhip_vec* y) {
    MGB_API_BEGIN_ON(y)
    same_ctx(x, y);
    MGB_REQUIRE(x->len == y->len, "axpy: length mismatch");
    if (y->len) launch_axpy(alpha, x->buf.p, y->buf.p, y->len, y->ctx->stream);
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_vec_scale(double alpha, mgbhip_vec* x) {
    MGB_API_BEGIN_ON(x)
    MGB_REQUIRE(x, "null vector");
    if (x->len) launch_scale_copy(x->buf.p, alpha, x->buf.p, x->len, x->ctx->stream);
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_vec_dot(const mgbhip_vec* a, const mgbhip_vec* b, double* out) {
    MGB_API_BEGIN_ON(a)
    same_ctx(a, b);
    MGB_REQUIRE(out && a->len == b->len, "dot: bad argument");
    *out = 0.0;
    if (a->len == 0) return MGBHIP_OK;
    mgbhip_ctx* c = a->ctx;
    double* sc = vec_scratch(c, a->len);
    launch_dot(a->buf.p, b->buf.p, a->len, sc, c->vscal.p, c->stream);
    c->vscal.download(out, 1, c->stream);
    MGB_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MGBHIP_OK;
    MGB_API_END
}

static int vec_stats_host(const mgbhip_vec* a, double* two) {
    mgbhip_ctx* c = a->ctx;
    two[0] = two[1] = 0.0;
    if (a->len == 0) return MGBHIP_OK;
    double* sc = vec_scratch(c, a->len);
    launch_vec_stats(a->buf.p, a->len, sc, c->vscal.p, c->stream);
    c->vscal.download(two, 2, c->stream);
    MGB_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MGBHIP_OK;
}

int mgbhip_vec_norm(const mgbhip_vec* a, double* out) {
    MGB_API_BEGIN_ON(a)
    MGB_REQUIRE(a && out, "null argument");
    double two[2];
    vec_stats_host(a, two);
    *out = std::sqrt(two[0]);
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_vec_isfinite(const mgbhip_vec* a, int32_t* all_finite) {
    MGB_API_BEGIN_ON(a)
    MGB_REQUIRE(a && all_finite, "null argument");
    double two[2];
    vec_stats_host(a, two);
    *all_finite = (two[1] == 0.0) ? 1 : 0;
    return MGBHIP_OK;
    MGB_API_END
}

static void check_closure_args(mgbhip_problem* P, int32_t level, const mgbhip_vec* s, const mgbhip_vec* c, const mgbhip_vec* z0) {
    check_level(P, level);
    MGB_REQUIRE(s && c && z0, "null vector");
    MGB_REQUIRE(s->ctx == P->ctx && c->ctx == P->ctx && z0->ctx == P->ctx, "vector and problem belong to different contexts");
    MGB_REQUIRE(s->len == P->levels[level].m, "s has the wrong length for this level");
    MGB_REQUIRE(c->len == P->n * P->nD, "c must hold n x nD entries");
    MGB_REQUIRE(z0->len == (int64_t)P->nu * P->n, "z0 must hold nu x n entries");
}

int mgbhip_f0_d(mgbhip_problem* P, int32_t level, const mgbhip_vec* s, const mgbhip_vec* c, const mgbhip_vec* z0, double* value) {
    MGB_API_BEGIN_ON(P)
    check_closure_args(P, level, s, c, z0);
    MGB_REQUIRE(value, "null argument");
    P->touch();                           // caller-owned vectors: no cached z0 + R*s can be trusted
    *value = P->eval_f0(level, s->buf.p, z0->buf.p, c->buf.p);
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_f1_d(mgbhip_problem* P, int32_t level, const mgbhip_vec* s, const mgbhip_vec* c, const mgbhip_vec* z0, mgbhip_vec* grad) {
    MGB_API_BEGIN_ON(P)
    check_closure_args(P, level, s, c, z0);
    MGB_REQUIRE(grad && grad->ctx == P->ctx && grad->len == s->len, "grad has the wrong length");
    P->touch();
    P->eval_f1(level, s->buf.p, z0->buf.p, c->buf.p, grad->buf.p);
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_f2_d(mgbhip_problem* P, int32_t level, const mgbhip_vec* s, const mgbhip_vec* c, const mgbhip_vec* z0) {
    MGB_API_BEGIN_ON(P)
    check_closure_args(P, level, s, c, z0);
    P->touch();
    P->eval_f2(level, s->buf.p, z0->buf.p, c->buf.p);
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_solve_d(mgbhip_problem* P, int32_t level, const mgbhip_vec* g, mgbhip_vec* x) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    MGB_REQUIRE(g && x && g->ctx == P->ctx && x->ctx == P->ctx, "bad vector");
    MGB_REQUIRE(g->len == P->levels[level].m && x->len == g->len, "solve: length mismatch");
    hipStream_t st = P->stream();
    P->factor(level);
    P->trisolve(level, g->buf.p, x->buf.p);
    const int status = P->levels[level].solver.status(st);
    if (status != MGBHIP_OK) g_last_error = "Cholesky met a non-positive pivot";
    return status;
    MGB_API_END
}

int mgbhip_prolong_add(mgbhip_problem* P, int32_t level, const mgbhip_vec* s, mgbhip_vec* z) {
    MGB_API_BEGIN_ON(P)
    check_level(P, level);
    MGB_REQUIRE(s && z && s->ctx == P->ctx && z->ctx == P->ctx, "bad vector");
    const Level& Lv = P->levels[level];
    MGB_REQUIRE(s->len == Lv.m && z->len == Lv.rows, "prolong_add: length mismatch");
    launch_csr_matvec(Lv.rows, Lv.Rptr.p, Lv.Rcol.p, Lv.Rval.p, s->buf.p, z->buf.p, true, false, P->stream());
    P->touch();
    return MGBHIP_OK;
    MGB_API_END
}

static int node_map(mgbhip_problem* P, const double* z, double* F, double* Dz, int mode) {
    MGB_REQUIRE(P && z && F, "null argument");
    hipStream_t st = P->stream();
    P->d_z0.upload(z, (size_t)P->nu * P->n, st);
    ElemParams E = P->base_params(-1, nullptr, P->d_z0.p, nullptr);
    if (Dz) {
        P->d_nodeDz.ensure((size_t)P->n * P->nD);
        E.out_Dz = P->d_nodeDz.p;
    }
    launch_elem(E, mode, st);
    P->d_nodeF.download(F, (size_t)P->n, st);
    if (Dz) P->d_nodeDz.download(Dz, (size_t)P->n * P->nD, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    return MGBHIP_OK;
}

int mgbhip_node_barrier(mgbhip_problem* P, const double* z, double* F, double* Dz) {
    MGB_API_BEGIN_ON(P)
    return node_map(P, z, F, Dz, MODE_NODE_F);
    MGB_API_END
}

int mgbhip_node_slack(mgbhip_problem* P, const double* z, double* slack) {
    MGB_API_BEGIN_ON(P)
    return node_map(P, z, slack, nullptr, MODE_NODE_SLACK);
    MGB_API_END
}

void mgbhip_default_options(mgbhip_options* o, int64_t n_nodes) {
    const double eps = std::numeric_limits<double>::epsilon();
    o->tol = std::sqrt(eps);
    o->t = 0.1;
    o->kappa = 10.0;
    o->maxit = 10000;
    o->max_newton = (int32_t)std::ceil(std::log2(-std::log2(eps)) + 2);
    o->ls_beta = 0.5;
    o->ls_c1 = 0.1;
    o->line_search = 0;
    o->stop_lambda_tol = 0.25 / std::sqrt((double)n_nodes);
    o->stop_theta = 0.9;
    o->finalize = 1;
    o->finalize_theta = 0.9;
    o->early_stop = 0;
    o->stopping_criterion = nullptr;
    o->early_stop_fn = nullptr;
    o->user = nullptr;
}

int mgbhip_mgb_core(mgbhip_problem* P, double* z, const double* c, const mgbhip_options* opt, mgbhip_core_result* res) {
    MGB_API_BEGIN_ON(P)
    MGB_REQUIRE(P && z && c && opt && res, "null argument");
    MGB_REQUIRE(opt->tol > 0 && opt->t > 0 && opt->kappa > 1 && opt->maxit >= 1 && opt->max_newton >= 1, "bad options");
    int rc = core_run(P, z, c, opt, res);
    if (rc == MGBHIP_ERR_CONVERGENCE)
        g_last_error = res->failure_code == 2 ? "Convergence failure in mgb_solve: iteration_limit"
                                              : "Convergence failure in mgb_solve: stall";
    return rc;
    MGB_API_END
}

int mgbhip_matched_t(mgbhip_problem* P, const double* z, const double* c, double t_default, double* t_out) {
    MGB_API_BEGIN_ON(P)
    MGB_REQUIRE(P && z && c && t_out, "null argument");
    return matched_t_run(P, z, c, t_default, t_out);
    MGB_API_END
}

int mgbhip_stage_ms(mgbhip_problem* P, const char* stage, double* total_ms, int64_t* launches) {
    MGB_API_BEGIN_ON(P)
    MGB_REQUIRE(P && stage, "null argument");
    P->ctx->timers.collect();
    auto it = P->ctx->timers.recs.find(stage);
    if (total_ms) *total_ms = it == P->ctx->timers.recs.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == P->ctx->timers.recs.end() ? 0 : it->second.launches;
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_solver_stats(mgbhip_problem* P, int32_t level, double* out) {
    MGB_API_BEGIN
    check_level(P, level);
    MGB_REQUIRE(out != nullptr, "null argument");
    const Level& L = P->levels[level];
    const MfPlan& pl = L.solver.plan;
    out[0] = (double)pl.fronts.size();
    out[1] = (double)pl.max_m;
    out[2] = (double)pl.arena_doubles;
    out[3] = (double)pl.factor_flops;
    out[4] = (double)pl.peeled;
    out[5] = pl.level_ptr.empty() ? 0.0 : (double)(pl.level_ptr.size() - 1);
    out[6] = (double)L.nnz;
    out[7] = (double)L.m;
    return MGBHIP_OK;
    MGB_API_END
}

int mgbhip_solver_chain(mgbhip_problem* P, int32_t level, double* out) {
    MGB_API_BEGIN
    check_level(P, level);
    MGB_REQUIRE(out != nullptr, "null argument");
    MGB_REQUIRE(P->levels[level].solver.analyzed, "mgbhip_solver_chain: the level has not been factored yet");
    P->levels[level].solver.chain_stats(out);
    return MGBHIP_OK;
    MGB_API_END
}

#ifdef MGB_STEP_PROBE
int mgbhip_debug_probe(long long* out64) { (void)hipDeviceSynchronize(); mgbhip::mf_debug_probe(out64); return 0; }
#endif

int mgbhip_reset_stage_timers(mgbhip_problem* P, int enable) {
    MGB_API_BEGIN_ON(P)
    MGB_REQUIRE(P, "null argument");
    P->ctx->timers.reset(enable != 0);
    return MGBHIP_OK;
    MGB_API_END
}

}  // extern "C"
