/*
 * mgbhip.h -- C ABI of libmgbhip.so, the MI355X (gfx950) backend for the inner
 * Newton hot path of MultiGridBarrier.jl.
 *
 * Boundary: the reference moves a CPU `MGBProblem` through
 * `native_to_device(D, prob)`, runs the barrier Newton loops on the device types,
 * and moves the `MGBSOL` back (reference: src/mgb.jl:798-842, src/device.jl:40-60).
 * This library is what a `HIPDevice <: Device` package extension binds by `ccall`:
 * plain pointers and sizes only, an opaque handle that owns all device memory,
 * assembly plans and factorizations (the reference keeps those in two process-global
 * caches flushed by `mgb_cleanup`, src/BlockMatrices.jl:320,737-751; here they die
 * with the handle).  INTEGRATION.md shows the Julia side.
 *
 * Conventions
 *  - every entry point returns an `int` status (MGBHIP_OK = 0); numerical
 *    infeasibility is NOT an error: barrier values come back as +Inf/NaN exactly
 *    like the reference's `Log` protocol (src/utils.jl:14, src/newton.jl:35-50);
 *  - all floating point is IEEE double; index arrays are 32-bit, 0-based;
 *  - host pointers unless a parameter is named `d_*`;
 *  - a handle is not thread-safe; distinct handles are independent; all work of a
 *    handle is issued on the stream given at creation (never the NULL stream
 *    implicitly), mirroring the stream discipline the reference's CUDA backend
 *    is tested for (test/test_cuda.jl:118-130).
 */
#ifndef MGBHIP_H
#define MGBHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGBHIP_OK 0
#define MGBHIP_ERR_INVALID 1      /* bad argument / unsupported functor family or size */
#define MGBHIP_ERR_HIP 2          /* a HIP runtime call failed (see mgbhip_last_error) */
#define MGBHIP_ERR_NOT_SPD 3      /* Cholesky met a non-positive pivot (reference: the
                                     Symmetric `\` would fall back / throw, src/utils.jl:145) */
#define MGBHIP_ERR_NONFINITE 4    /* a Newton precondition failed (src/newton.jl:238-254) */
#define MGBHIP_ERR_CONVERGENCE 5  /* MGBConvergenceFailure (src/utils.jl:178-184); code in diagnostics */

#define MGBHIP_MAX_PIECES 4
#define MGBHIP_MAX_IDX 4
#define MGBHIP_MAX_ND 10   /* 3-D parabolic phase I: (dim + 3) + 1 + 3 rows */
#define MGBHIP_MAX_NU 4
#define MGBHIP_MAX_OPS 8

#define MGBHIP_KIND_EP 1          /* convex_Euclidian_power (src/convex_euclidian_power.jl:352-453) */
#define MGBHIP_KIND_LINEAR 2      /* convex_linear          (src/convex_linear.jl:78-223)           */

typedef struct mgbhip_ctx mgbhip_ctx;          /* device + stream + workspace            */
typedef struct mgbhip_problem mgbhip_problem;  /* one AMG (src/multigrid.jl:278-288) + one Convex */

/* One piece of a Convex (src/convex.jl:80-86).  Grids are n x K, column-major, as the
 * reference's `Q.args`; NULL selects the documented default without reading memory. */
typedef struct {
    int32_t kind;                 /* MGBHIP_KIND_*                                        */
    int32_t ni;                   /* length of idx (EP: nz)                               */
    int32_t nc;                   /* LINEAR: constraint rows; EP: ignored (= ni)          */
    int32_t idx[MGBHIP_MAX_IDX];  /* 0-based positions into y                             */
    const double* A;              /* n x (nc*ni), per-node matrix column-major; NULL = I  */
    const double* b;              /* n x nc; NULL = 0                                     */
    const double* p;              /* EP: n; NULL = use p_const / mu_const                 */
    const double* mu;             /* EP: n (src/convex_euclidian_power.jl:380-381)        */
    double p_const, mu_const;
    const double* select;         /* n: non-zero = piece active at the node; NULL = all   */
} mgbhip_piece;

typedef struct {
    int32_t npieces;
    mgbhip_piece pieces[MGBHIP_MAX_PIECES];
    /* phase-I wrapper `_feasibility_convex` (src/mgb.jl:217-287): when `feasibility`
     * is non-zero the node barrier is cobarrier(y[0:NC]) plus the box terms, with
     * NC = nD_main + 1; b and R are set per box round by mgbhip_problem_set_box.   */
    int32_t feasibility;
    int32_t NC;
} mgbhip_cone;

/* CSR image of one prolongation R_fine[l] (src/multigrid.jl:491): (nu*n) x m. */
typedef struct {
    int64_t rows, cols;
    const int32_t* rowptr;        /* rows + 1 */
    const int32_t* colidx;
    const double* values;
} mgbhip_csr;

typedef struct {
    int32_t p;                    /* nodes per element (block size)                       */
    int64_t N;                    /* elements; n = p*N broken nodes                       */
    int32_t nu;                   /* state components                                     */
    int32_t nD;                   /* rows of D                                            */
    int32_t n_ops;                /* distinct operator arrays                             */
    const double* ops[MGBHIP_MAX_OPS]; /* each p x p x N, Julia Array{T,3} layout; NULL = identity
                                     (BlockDiag, src/BlockMatrices.jl:17-22)              */
    int32_t D_state[MGBHIP_MAX_ND];    /* state component of D row k (BlockColumn.active_col) */
    int32_t D_op[MGBHIP_MAX_ND];       /* operator index of D row k                        */
    const double* w;              /* n quadrature weights                                 */
    int32_t L;                    /* hierarchy depth                                      */
    const mgbhip_csr* R;          /* L prolongations, coarsest first                      */
    mgbhip_cone cone;
    const double* barrier_weights;/* n, or NULL for the flat (1/n) average (src/convex.jl:279-304) */
    /* Node coordinates `AMG.x` (src/multigrid.jl:280), n x dim column-major, or NULL.  Used only as an
     * ordering hint by the sparse direct solve (geometric nested dissection on the centroids of the
     * level-J basis functions); NULL falls back to a graph-only dissection.  Results do not depend
     * on it beyond rounding. */
    const double* x;
    int32_t dim;
} mgbhip_problem_desc;

/* Solver controls (reference defaults: src/mgb.jl:95-101, :360-363, src/newton.jl:139). */
typedef struct {
    double tol;                   /* sqrt(eps)            */
    double t;                     /* 0.1                  */
    double kappa;                 /* 10                   */
    int32_t maxit;                /* 10000                */
    int32_t max_newton;           /* ceil(log2(-log2 eps)) + 2 = 8 */
    double ls_beta, ls_c1;        /* backtracking 0.5, 0.1 */
    int32_t line_search;          /* 0 backtracking, 1 illinois */
    double stop_lambda_tol;       /* stopping_inexact(0.25/sqrt(n), 0.9); <0 => stopping_exact(stop_theta) */
    double stop_theta;
    int32_t finalize;             /* 1: stopping_exact(finalize_theta); 0: NoFinalize */
    double finalize_theta;        /* 0.9 */
    int32_t early_stop;           /* 0 none; 1 phase-I margin rule (src/mgb.jl:486-491)   */
    /* Optional user callables, NULL = the built-in rule selected above.  They receive only what the
     * reference hands to the corresponding Julia callable:
     *   stopping_criterion(ymin, ynext, gmin, |gnext|, sqrt(incmin), sqrt(inc)): the reference's
     *   `stop(ymin, ynext, gmin, gnext, n, ndecmin, ndec)` (src/newton.jl:187,222-225,279; kwarg of mgb_solve,
     *   src/mgb.jl:360) with the vectors gnext, n reduced to the norm both built-in rules use -> non-zero = converged;
     *   early_stop(z, t): z is the current fine iterate copied to the host (nu*n doubles), called
     *   between completed t-steps like `early_stop(z)` (src/mgb.jl:85-89,138) -> non-zero = stop.  */
    int (*stopping_criterion)(double ymin, double ynext, double gmin, double gnorm_next, double ndecmin, double ndec,
                              void* user);
    int (*early_stop_fn)(const double* z, double t, void* user);
    void* user;
} mgbhip_options;

/* Diagnostics of one mgb_core run (the fields of SOL_main, src/mgb.jl:176-182). */
typedef struct {
    int32_t k;                    /* t-steps taken                                         */
    int32_t L;
    int32_t failure_code;         /* 0 ok, 1 :stall, 2 :iteration_limit                    */
    double t_final, t_elapsed;
    double solve_seconds;         /* wall time inside factor+solve (reported separately)   */
    int64_t newton_iterations;    /* sum(its)                                              */
    int64_t f0_evals, f1_evals, f2_evals, factorizations;
    /* caller-provided, capacity in cap_steps: its is L x cap_steps column-major */
    int32_t cap_steps;
    int64_t* its;
    double* ts;
    double* kappas;
    double* times;
    double* c_dot_Dz;
} mgbhip_core_result;

/* ---- lifecycle --------------------------------------------------------------------- */
int mgbhip_create(mgbhip_ctx** ctx, int device_id, void* hip_stream /* NULL: private stream */);
int mgbhip_destroy(mgbhip_ctx* ctx);
const char* mgbhip_last_error(void);
const char* mgbhip_version(void);

/* native_to_device for one (AMG, Convex) pair (ext/MultiGridBarrierCUDAExt/conversion.jl:152-159).
 * `share` may name an existing problem of the same ctx whose operator arrays and weights
 * are identical (the (main, feasibility) pair shares them; test/test_cuda.jl:80-99). */
int mgbhip_problem_create(mgbhip_ctx* ctx, const mgbhip_problem_desc* desc,
                          mgbhip_problem* share, mgbhip_problem** out);
int mgbhip_problem_destroy(mgbhip_problem* prob);      /* also flushes plans + factorizations */
int mgbhip_problem_set_box(mgbhip_problem* prob, double b, double R);
int mgbhip_problem_set_barrier_weights(mgbhip_problem* prob, const double* bw /* n or NULL */);
int64_t mgbhip_level_size(const mgbhip_problem* prob, int32_t level);   /* m_J = ncols R_fine[J] */

/* ---- the Barrier closures (src/convex.jl:155-202) at level J (0-based), host vectors ----
 * s: m_J, c: n x nD column-major (= t * f_grid), z0: nu*n.                              */
int mgbhip_f0(mgbhip_problem* prob, int32_t level, const double* s, const double* c,
              const double* z0, double* value);
int mgbhip_f1(mgbhip_problem* prob, int32_t level, const double* s, const double* c,
              const double* z0, double* grad /* m_J */);
/* f2 assembles H = R' H_blk R on the device and optionally copies it out as CSR
 * (pattern fixed per level: query with mgbhip_hessian_pattern).                          */
int mgbhip_f2(mgbhip_problem* prob, int32_t level, const double* s, const double* c,
              const double* z0, double* values /* nnz or NULL */);
int mgbhip_hessian_pattern(mgbhip_problem* prob, int32_t level, int64_t* nnz,
                           const int32_t** rowptr, const int32_t** colidx);
/* n = solve(symmetric(H), g) with the H of the last mgbhip_f2 at this level
 * (src/newton.jl:253, src/utils.jl:142-145): sparse Cholesky on the device.              */
int mgbhip_solve(mgbhip_problem* prob, int32_t level, const double* g, double* x);
/* The reference's `solve(A, b)` hook (src/utils.jl:142-145; cuDSS twin cudss_solver.jl:396-408) takes any
 * matrix with the level's sparsity pattern: replace the values of H (CSR order of mgbhip_hessian_pattern);
 * the next mgbhip_solve / mgbhip_solve_newton factors them.                                                */
int mgbhip_set_hessian(mgbhip_problem* prob, int32_t level, const double* values /* nnz */);
/* The solve exactly as the resident Newton loop performs it (src/newton.jl:253-255): the bordered matrix
 * [H -g; -g' -1] is factored, so the forward substitution rides along the factorization, and one backward
 * sweep returns x = H^{-1} g; lambda2 (optional) = <g, x>.                                                  */
int mgbhip_solve_newton(mgbhip_problem* prob, int32_t level, const double* g, double* x, double* lambda2);
/* ---- one process per GPU: domain decomposition of the resident loop ----
 * The problem handed to mgbhip_problem_create is then this rank's SLICE: its elements, and per level the columns of
 * R restricted to the unknowns its elements touch plus the interface unknowns (those whose support meets more than
 * one rank), both in an order common to all ranks.  Per level: the interface columns (local indices, ascending) and
 * an ownership mask (1 where this rank counts an unknown in dot products and norms, 0 elsewhere; interface unknowns
 * are owned by exactly one rank).  The library then keeps s distributed with a replicated interface: f0 and the
 * scalars of the Newton loop are summed over ranks, f1 sums its interface entries, every rank eliminates its
 * interior unknowns and the assembled interface front of the factorization is summed over ranks before each rank
 * factors it.  `allreduce` sums (op 0) or maximises (op 1) `count` doubles in place over all ranks and returns 0;
 * the buffer is host memory unless accepts_device_ptr was set, in which case large buffers are passed as device
 * pointers that are ready on the handle's stream when the call is made (the callee must complete before it returns).
 * Must be set before the first evaluation; mgbhip_mgb_core then runs the same control flow on every rank.       */
typedef int (*mgbhip_allreduce_fn)(void* user, double* buf, int64_t count, int32_t op, int32_t on_device);
int mgbhip_problem_set_sharding(mgbhip_problem* prob, int32_t level, int64_t n_iface,
                                const int32_t* iface_cols, const double* own_mask /* m_J */);
int mgbhip_problem_set_collective(mgbhip_problem* prob, mgbhip_allreduce_fn allreduce, void* user,
                                  int32_t accepts_device_ptr);
/* One Newton direction exactly as the resident loop forms it at (s, c, z0): g = f1, H = f2 left in the
 * element-block slab (fine levels: leaf fronts condensed inside the element kernel from the second call on),
 * bordered factorization, backward sweep.  x = H^{-1} g, lambda2 = <g, x>; *condensed (optional) reports
 * whether the element kernel wrote the leaf fronts.  Test / measurement hook.                               */
int mgbhip_newton_direction(mgbhip_problem* prob, int32_t level, const double* s, const double* c,
                            const double* z0, double* x, double* lambda2, int32_t* condensed);
/* Per-node barrier value map_rows_gpu(F0, args..., Dz(z)) (src/mgb.jl:410-420) and the
 * slack initialiser (src/mgb.jl:437-440); y is n x nD column-major.                      */
int mgbhip_node_barrier(mgbhip_problem* prob, const double* z, double* F /* n */, double* Dz /* n*nD or NULL */);
int mgbhip_node_slack(mgbhip_problem* prob, const double* z, double* slack /* n */);

/* ---- the loops (src/newton.jl:227-287, src/mgb.jl:16-183), resident on the device ---- */
int mgbhip_mgb_core(mgbhip_problem* prob, double* z /* nu*n in/out */, const double* c /* n x nD */,
                    const mgbhip_options* opt, mgbhip_core_result* res);
/* _matched_t (src/mgb.jl:307-330) */
int mgbhip_matched_t(mgbhip_problem* prob, const double* z, const double* c, double t_default,
                     double* t_out);
void mgbhip_default_options(mgbhip_options* opt, int64_t n_nodes);

/* ---- device-resident vectors and closures ------------------------------------------------------
 * The "fine" integration style of SURVEY.md section 8b: a binding that keeps the reference's generic
 * `newton` / `mgb_step` (src/newton.jl:227-287, src/mgb.jl:16-82) wraps `mgbhip_vec` in its device
 * vector type; nothing crosses PCIe per call except scalars.  A vector belongs to one context and is
 * used on that context's stream.  (The CUDA extension gets these from CuArray broadcasting:
 * ext/MultiGridBarrierCUDAExt/mgb_interface.jl:14-41.)                                           */
typedef struct mgbhip_vec mgbhip_vec;
int mgbhip_vec_alloc(mgbhip_ctx* ctx, int64_t len, mgbhip_vec** out);      /* mgb_zeros: zero-filled  */
int mgbhip_vec_free(mgbhip_vec* v);
int64_t mgbhip_vec_len(const mgbhip_vec* v);
int mgbhip_vec_upload(mgbhip_vec* v, const double* host, int64_t len);
int mgbhip_vec_download(const mgbhip_vec* v, double* host, int64_t len);
int mgbhip_vec_fill(mgbhip_vec* v, double value);
int mgbhip_vec_copy(mgbhip_vec* dst, const mgbhip_vec* src);
int mgbhip_vec_axpy(double alpha, const mgbhip_vec* x, mgbhip_vec* y);      /* y += alpha x            */
int mgbhip_vec_scale(double alpha, mgbhip_vec* x);                          /* x *= alpha              */
int mgbhip_vec_dot(const mgbhip_vec* a, const mgbhip_vec* b, double* out);
int mgbhip_vec_norm(const mgbhip_vec* a, double* out);                      /* 2-norm                  */
int mgbhip_vec_isfinite(const mgbhip_vec* a, int32_t* all_finite);          /* mgb_all_isfinite        */
/* The Barrier closures and the direct solve on device vectors (s: m_J, c: n*nD column-major,
 * z0: nu*n, grad/x: m_J); f2_d leaves H assembled on the device for solve_d.                     */
int mgbhip_f0_d(mgbhip_problem* prob, int32_t level, const mgbhip_vec* s, const mgbhip_vec* c,
                const mgbhip_vec* z0, double* value);
int mgbhip_f1_d(mgbhip_problem* prob, int32_t level, const mgbhip_vec* s, const mgbhip_vec* c,
                const mgbhip_vec* z0, mgbhip_vec* grad);
int mgbhip_f2_d(mgbhip_problem* prob, int32_t level, const mgbhip_vec* s, const mgbhip_vec* c,
                const mgbhip_vec* z0);
int mgbhip_solve_d(mgbhip_problem* prob, int32_t level, const mgbhip_vec* g, mgbhip_vec* x);
/* z += R_J s  (src/mgb.jl:60) */
int mgbhip_prolong_add(mgbhip_problem* prob, int32_t level, const mgbhip_vec* s, mgbhip_vec* z);

/* ---- measurement hooks (bench.py): device-time of the named stage of the last call,
 * from hipEvents on the handle's stream.                                               */
int mgbhip_stage_ms(mgbhip_problem* prob, const char* stage, double* total_ms, int64_t* launches);
int mgbhip_reset_stage_timers(mgbhip_problem* prob, int enable);
/* Factorization statistics of a level (after its first solve): out[0] fronts, [1] largest
 * front, [2] arena doubles, [3] factor flops, [4] peeled unknowns, [5] tree levels,
 * [6] nnz(H), [7] unknowns.                                                             */
int mgbhip_solver_stats(mgbhip_problem* prob, int32_t level, double* out8);
/* Shape of one factorization + backward sweep of a level as the device runs it (bench.py: roofline_solver): out[0]
 * sequential 32-column pivot blocks on the critical path of the large fronts, [1] tree levels on the large-front
 * path, [2] kernel launches per factorization, [3] per backward sweep, [4] arena doubles, [5] factor flops,
 * [6] doubles the trailing updates move beyond one pass over the arena, [7] reserved.  The reference's counterpart is
 * opaque (cuDSS FACTORIZATION + SOLVE per Newton iteration, ext/MultiGridBarrierCUDAExt/cudss_solver.jl:279-288). */
int mgbhip_solver_chain(mgbhip_problem* prob, int32_t level, double* out8);

#ifdef __cplusplus
}
#endif
#endif /* MGBHIP_H */
