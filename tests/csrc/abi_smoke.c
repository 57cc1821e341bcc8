/*
 * abi_smoke.c -- a compiled C caller of include/mgbhip.h (no Python, no ctypes mirror): reads a
 * problem image written by tests/test_c_abi.py, builds the descriptor exactly as a foreign-language
 * binding would, runs the reference's default solve through mgbhip_mgb_core and
 * prints the solution; then repeats one Newton step with the device-vector entry points
 * (mgbhip_vec_*, mgbhip_f0_d/f1_d/f2_d/solve_d/prolong_add) to check that the "fine" integration style
 * of SURVEY.md section 8b gives the same numbers as the host-vector calls.
 *
 * File format (little endian): int32 p, int64 N, int32 nu, nD, n_ops, L; per D row int32 state, op;
 * per op: int32 identity, then (if not) p*p*N doubles; w (n doubles); x: int32 dim, n*dim doubles;
 * per level: int64 rows, cols, nnz, rowptr (rows+1 int32), colidx (nnz int32), values (nnz doubles);
 * cone: int32 ni, idx[ni] int32, double p_const, mu_const;  f grid (n*nD doubles, column-major);
 * g stacked (nu*n doubles).
 *
 * Build (done by __graft_entry__.build()):
 *   gcc -std=c99 -O2 -Iinclude tests/csrc/abi_smoke.c -o tests/csrc/abi_smoke \
 *       -Lmultigridbarrier.jl_amd/lib -lmgbhip -Wl,-rpath,$PWD/multigridbarrier.jl_amd/lib -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mgbhip.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int _rc = (call);                                                                \
        if (_rc != MGBHIP_OK) {                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #call, _rc, mgbhip_last_error());        \
            return 10 + _rc;                                                             \
        }                                                                                \
    } while (0)

static void* rd(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "short read\n");
        exit(3);
    }
    return p;
}
static int32_t rd_i32(FILE* f) { int32_t v; if (fread(&v, 4, 1, f) != 1) exit(3); return v; }
static int64_t rd_i64(FILE* f) { int64_t v; if (fread(&v, 8, 1, f) != 1) exit(3); return v; }
static double rd_f64(FILE* f) { double v; if (fread(&v, 8, 1, f) != 1) exit(3); return v; }

#include <stddef.h>
#define SZ(T) printf("sizeof " #T " %zu\n", sizeof(T))
#define OFF(T, f) printf("offsetof " #T "." #f " %zu\n", offsetof(T, f))

/* `abi_smoke --layout`: the compiler's view of every struct in the header, for the ctypes mirror test */
static int print_layout(void) {
    SZ(mgbhip_piece); OFF(mgbhip_piece, idx); OFF(mgbhip_piece, A); OFF(mgbhip_piece, p_const); OFF(mgbhip_piece, select);
    SZ(mgbhip_cone); OFF(mgbhip_cone, pieces); OFF(mgbhip_cone, feasibility); OFF(mgbhip_cone, NC);
    SZ(mgbhip_csr); OFF(mgbhip_csr, rowptr); OFF(mgbhip_csr, values);
    SZ(mgbhip_problem_desc); OFF(mgbhip_problem_desc, N); OFF(mgbhip_problem_desc, ops); OFF(mgbhip_problem_desc, D_state);
    OFF(mgbhip_problem_desc, w); OFF(mgbhip_problem_desc, R); OFF(mgbhip_problem_desc, cone);
    OFF(mgbhip_problem_desc, barrier_weights); OFF(mgbhip_problem_desc, x); OFF(mgbhip_problem_desc, dim);
    SZ(mgbhip_options); OFF(mgbhip_options, maxit); OFF(mgbhip_options, ls_beta); OFF(mgbhip_options, line_search);
    OFF(mgbhip_options, stop_lambda_tol); OFF(mgbhip_options, finalize); OFF(mgbhip_options, finalize_theta);
    OFF(mgbhip_options, early_stop); OFF(mgbhip_options, stopping_criterion); OFF(mgbhip_options, early_stop_fn);
    OFF(mgbhip_options, user);
    SZ(mgbhip_core_result); OFF(mgbhip_core_result, failure_code); OFF(mgbhip_core_result, t_final);
    OFF(mgbhip_core_result, newton_iterations); OFF(mgbhip_core_result, cap_steps); OFF(mgbhip_core_result, its);
    OFF(mgbhip_core_result, c_dot_Dz);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: abi_smoke problem.bin | --layout\n");
        return 2;
    }
    if (strcmp(argv[1], "--layout") == 0) return print_layout();
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    mgbhip_problem_desc d;
    memset(&d, 0, sizeof d);
    d.p = rd_i32(f);
    d.N = rd_i64(f);
    d.nu = rd_i32(f);
    d.nD = rd_i32(f);
    d.n_ops = rd_i32(f);
    d.L = rd_i32(f);
    const int64_t n = (int64_t)d.p * d.N;
    for (int k = 0; k < d.nD; ++k) { d.D_state[k] = rd_i32(f); d.D_op[k] = rd_i32(f); }
    for (int o = 0; o < d.n_ops; ++o) {
        const int32_t ident = rd_i32(f);
        d.ops[o] = ident ? NULL : (const double*)rd(f, sizeof(double) * (size_t)d.p * d.p * d.N);
    }
    d.w = (const double*)rd(f, sizeof(double) * (size_t)n);
    d.dim = rd_i32(f);
    d.x = (const double*)rd(f, sizeof(double) * (size_t)n * d.dim);
    mgbhip_csr* R = (mgbhip_csr*)calloc((size_t)d.L, sizeof(mgbhip_csr));
    for (int l = 0; l < d.L; ++l) {
        R[l].rows = rd_i64(f);
        R[l].cols = rd_i64(f);
        const int64_t nnz = rd_i64(f);
        R[l].rowptr = (const int32_t*)rd(f, 4 * (size_t)(R[l].rows + 1));
        R[l].colidx = (const int32_t*)rd(f, 4 * (size_t)nnz);
        R[l].values = (const double*)rd(f, 8 * (size_t)nnz);
    }
    d.R = R;
    d.cone.npieces = 1;
    mgbhip_piece* pc = &d.cone.pieces[0];
    pc->kind = MGBHIP_KIND_EP;
    pc->ni = rd_i32(f);
    pc->nc = pc->ni;
    for (int i = 0; i < pc->ni; ++i) pc->idx[i] = rd_i32(f);
    pc->p_const = rd_f64(f);
    pc->mu_const = rd_f64(f);
    double* fgrid = (double*)rd(f, 8 * (size_t)n * d.nD);
    double* z = (double*)rd(f, 8 * (size_t)n * d.nu);
    fclose(f);

    mgbhip_ctx* ctx = NULL;
    mgbhip_problem* P = NULL;
    CHECK(mgbhip_create(&ctx, 0, NULL));
    CHECK(mgbhip_problem_create(ctx, &d, NULL, &P));

    /* ---- one Newton step at the fine level, host vectors vs device vectors -------------------- */
    const int32_t J = d.L - 1;
    const int64_t m = mgbhip_level_size(P, J);
    double* c = (double*)malloc(8 * (size_t)n * d.nD);
    for (int64_t i = 0; i < n * d.nD; ++i) c[i] = 0.1 * fgrid[i];
    double* s = (double*)calloc((size_t)m, 8);
    double* g = (double*)malloc(8 * (size_t)m);
    double* x = (double*)malloc(8 * (size_t)m);
    double y0 = 0;
    CHECK(mgbhip_f0(P, J, s, c, z, &y0));
    CHECK(mgbhip_f1(P, J, s, c, z, g));
    CHECK(mgbhip_f2(P, J, s, c, z, NULL));
    CHECK(mgbhip_solve(P, J, g, x));
    double inc = 0;
    for (int64_t i = 0; i < m; ++i) inc += g[i] * x[i];

    mgbhip_vec *vs, *vc, *vz, *vg, *vx;
    CHECK(mgbhip_vec_alloc(ctx, m, &vs));
    CHECK(mgbhip_vec_alloc(ctx, n * d.nD, &vc));
    CHECK(mgbhip_vec_alloc(ctx, n * d.nu, &vz));
    CHECK(mgbhip_vec_alloc(ctx, m, &vg));
    CHECK(mgbhip_vec_alloc(ctx, m, &vx));
    CHECK(mgbhip_vec_upload(vc, fgrid, n * d.nD));
    CHECK(mgbhip_vec_scale(0.1, vc));
    CHECK(mgbhip_vec_upload(vz, z, n * d.nu));
    double y0d = 0, incd = 0, nrm = 0;
    int32_t fin = 0;
    CHECK(mgbhip_f0_d(P, J, vs, vc, vz, &y0d));
    CHECK(mgbhip_f1_d(P, J, vs, vc, vz, vg));
    CHECK(mgbhip_f2_d(P, J, vs, vc, vz));
    CHECK(mgbhip_solve_d(P, J, vg, vx));
    CHECK(mgbhip_vec_dot(vg, vx, &incd));
    CHECK(mgbhip_vec_norm(vg, &nrm));
    CHECK(mgbhip_vec_isfinite(vx, &fin));
    /* backtracking line search on the device vectors (src/newton.jl:139-154): s = -sigma x, sigma halved
     * until the objective is finite and satisfies Armijo; then z += R s on the device */
    double y1 = 0, sigma = 1.0;
    for (int it = 0; it < 60; ++it, sigma *= 0.5) {
        CHECK(mgbhip_vec_fill(vs, 0.0));
        CHECK(mgbhip_vec_axpy(-sigma, vx, vs));
        CHECK(mgbhip_f0_d(P, J, vs, vc, vz, &y1));
        if (isfinite(y1) && y1 <= y0d - 0.1 * sigma * incd) break;
    }
    CHECK(mgbhip_prolong_add(P, J, vs, vz));
    CHECK(mgbhip_vec_fill(vs, 0.0));
    double y1b = 0;
    CHECK(mgbhip_f0_d(P, J, vs, vc, vz, &y1b));
    double gn = 0;
    for (int64_t i = 0; i < m; ++i) gn += g[i] * g[i];
    printf("newton_step host %.17g %.17g device %.17g %.17g norm %.17g %.17g finite %d armijo %.17g %.17g %.17g\n", y0, inc,
           y0d, incd, sqrt(gn), nrm, (int)fin, y1, y1b, y0 - 0.1 * sigma * inc);
    mgbhip_vec_free(vs); mgbhip_vec_free(vc); mgbhip_vec_free(vz); mgbhip_vec_free(vg); mgbhip_vec_free(vx);

    /* ---- the full solve: mgb_core with the reference defaults ----------------------------------- */
    mgbhip_options opt;
    mgbhip_default_options(&opt, n);
    mgbhip_core_result res;
    memset(&res, 0, sizeof res);
    enum { CAP = 256 };
    res.cap_steps = CAP;
    res.its = (int64_t*)calloc((size_t)CAP * d.L, 8);
    res.ts = (double*)calloc(CAP, 8);
    res.kappas = (double*)calloc(CAP, 8);
    res.times = (double*)calloc(CAP, 8);
    res.c_dot_Dz = (double*)calloc(CAP, 8);
    CHECK(mgbhip_mgb_core(P, z, fgrid, &opt, &res));
    printf("core k %d newton %lld t_final %.17g\n", (int)res.k, (long long)res.newton_iterations, res.t_final);
    printf("z");
    for (int64_t i = 0; i < n * d.nu; ++i) printf(" %.17g", z[i]);
    printf("\n");
    CHECK(mgbhip_problem_destroy(P));
    CHECK(mgbhip_destroy(ctx));
    return 0;
}
