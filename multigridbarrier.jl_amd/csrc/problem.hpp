// problem.hpp -- device-resident image of one (AMG, Convex) pair and its per-level plans.
#pragma once
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mgbhip.h"
#include "common.hpp"
#include "kernels.hpp"
#include "mf_solver.hpp"

struct mgbhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    mgbhip::StageTimers timers;
    mgbhip::DevBuf<double> vscratch, vscal;   // reduction scratch of the device-vector API
};

struct mgbhip_vec {
    mgbhip_ctx* ctx = nullptr;
    mgbhip::DevBuf<double> buf;
    int64_t len = 0;
};

namespace mgbhip {

// Operator arrays + weights, shared between the (main, feasibility) problems of one solve
// (the reference shares them by identity-memoised conversion, conversion.jl:1-11).
struct OpStore {
    std::vector<DevBuf<double>> ops;
    std::vector<bool> identity;
    DevBuf<double> w;
};

struct Level {
    int64_t rows = 0, m = 0;
    std::vector<int32_t> hRptr, hRcol;     // host copy of R (plan construction)
    std::vector<double> hRval;
    DevBuf<int32_t> Rptr, Rcol, Tptr, Tcol;
    DevBuf<double> Rval, Tval;
    bool T_long = false, R_long = false;
    bool R_unit = false;                  // every row of R: at most one entry, equal to 1 -> the element kernels prolong (Rsel)
    DevBuf<int32_t> Rsel;
    int32_t T_chunks = 0;                 // > 0: rows of R' are long enough for the chunked matvec
    // assembly plan for H = R' H_blk R (reference: BlockAssemblyPlan, src/BlockMatrices.jl:281-491)
    bool planned = false;
    bool selection = false;               // every row of R has at most one entry, equal to 1
    bool acc = false;                     // small coarse level: dense H from per-wave accumulators
    int32_t acc_waves = 0, acc_ctmax = 0;  // acc_waves: element streams
    int32_t acc_split = 1, acc_chunk = 0;  // chunks of the packed upper triangle, one per workgroup column
    DevBuf<double> acc_copies;
    std::vector<int32_t> hHptr, hHcol;
    DevBuf<int32_t> Hptr, Hcol, cptr, cidx, ecol_ptr, ecols, eoff;
    DevBuf<int32_t> spos;                 // projected levels: slab position of (element block entry) = its place in its contribution list
    bool sorted_slab = false;             // the projection kernels scatter through spos and the gather streams contiguous runs
    DevBuf<double> Hval, panels, slab;
    int64_t slab_doubles = 0;
    int32_t cmax = 1;
    bool long_lists = false;
    int32_t gather_chunk = 0, gather_nchunk = 0;   // very long lists: two-stage gather (chunk length, chunks per list)
    DevBuf<double> gather_part;
    DevBuf<int32_t> upq;                  // general (projected) levels: CSR positions with col >= row, what the Newton loop assembles
    int64_t nup = 0;
    int64_t nnz = 0;
    // dense (spectral) levels: DR = [D_k R_{state(k)}]_k stacked ((nD*n) x m) and W = Ybar * DR
    DevBuf<double> denseDR, denseW;
    MfSolver solver;
    bool have_H = false, factored = false;
    // direct values (fine / selection levels): the solver reads single-contribution entries of H straight from the
    // element-block slab d_hel and the shared ones from a compact array behind it: no CSR value array is formed
    bool direct = false;
    int64_t nshared = 0;
    DevBuf<int32_t> sh_q;                 // CSR positions of the shared entries
    std::vector<int32_t> h_vmap;          // CSR position -> index into [slab | shared | border]
    bool H_in_slab = false;               // the last eval_f2 of this level left H in d_hel (not in Hval)
    // domain decomposition (mgbhip_problem_set_sharding): interface columns and the ownership mask of this rank
    bool sharded = false;
    std::vector<int32_t> h_iface;
    DevBuf<int32_t> d_iface;
    DevBuf<double> own, iface_buf;
    bool condense_tried = false, condense = false;   // leaf fronts written by the element kernel (kernels.hpp)
    bool H_condensed = false;             // the last eval_f2 did so: d_hel holds no blocks, the arena holds the leaves
    const double* condensed_rhs = nullptr;
    int border_state2 = 0;                // 2: the current factors came from the slab path with a Newton right-hand side
    int border_state = 0;                 // tail of Hval: 0 unset, 1 identity border (solve), 2 Newton right-hand side (solve_border)
};

struct Counters {
    int64_t f0 = 0, f1 = 0, f2 = 0, factor = 0, newton = 0;
    double solve_seconds = 0;
};

}  // namespace mgbhip

struct mgbhip_problem {
    mgbhip_ctx* ctx = nullptr;
    int32_t p = 0, nu = 0, nD = 0;
    int64_t N = 0, n = 0;
    int32_t diag_mask_sel = 0;             // element blocks that are diagonal (identity-only states), compact on selection levels
    bool dense = false;                    // one dense spectral element (p > 64): dense.hip path
    std::shared_ptr<mgbhip::OpStore> store;
    int32_t D_state[MGBHIP_MAX_ND], D_op[MGBHIP_MAX_ND], D_stage[MGBHIP_MAX_ND];
    int32_t nstage = 0;
    const double* stage_ptr[MGBHIP_MAX_OPS];
    mgbhip::ConeDev cone;
    std::vector<mgbhip::DevBuf<double>> cone_grids;
    mgbhip::DevBuf<double> bw;
    bool has_bw = false;
    std::vector<mgbhip::Level> levels;
    std::vector<double> hx;                // host copy of the node coordinates (n x dim, column-major) or empty
    int32_t xdim = 0;
    // workspace
    mgbhip::DevBuf<double> d_z, d_z0, d_zfull, d_c, d_ret, d_hel, d_partials, d_scal, d_scratch, d_nodeF, d_nodeDz, d_dnDz, d_dnY, d_tchunk;
    mgbhip::DevBuf<double> d_x, d_g, d_nv, d_xn, d_gn, d_tmp, d_c0;
    mgbhip::DevBuf<int32_t> d_flag;
    int32_t step_stamp = 0;                // launch_step writes ++step_stamp into d_flag[0] when the step moved the iterate
    mgbhip::PinnedBuf pin;                 // scalar read-backs of the Newton loop
    mgbhip::Counters cnt;
    // z0 + R*s is cached in d_zfull across the f0/f1/f2 calls at one point: the key is the
    // (level, s, z) pointers plus a stamp every writer of those vectors bumps (touch()).
    int64_t zstamp = 0;
    mutable int64_t zf_stamp = -1;
    mutable int zf_level = -1;
    mutable const double* zf_s = nullptr;
    mutable const double* zf_z = nullptr;
    void touch() { ++zstamp; }
    // one process per GPU (include/mgbhip.h: mgbhip_problem_set_collective)
    mgbhip_allreduce_fn coll_fn = nullptr;
    void* coll_user = nullptr;
    bool coll_device = false;
    bool sharded() const { return coll_fn != nullptr; }
    std::vector<double> coll_host;
    void allreduce_host(double* h, int64_t count, int op);          // in place
    void allreduce_device(double* d, int64_t count, int op);        // in place, ordered on the handle's stream
    // g (m_J entries, this rank's partial sums) -> interface entries summed over ranks; d_part (optional) keeps the partial
    void reduce_interface(int level, double* d_g, double* d_part);
    const double* own_mask(int level) const { return levels[level].sharded ? levels[level].own.p : nullptr; }
    mgbhip::DevBuf<double> d_gpart, d_gnpart;                        // partial gradients (right-hand sides of the local eliminations)

    mgbhip::ElemParams base_params(int level, const double* d_s, const double* d_zz, const double* d_cc) const;
    hipStream_t stream() const { return ctx->stream; }
    void ensure_plan(int level);
    void ensure_analysis(int level);
    void ensure_direct(int level);
    void prepare_all();
    // Read-backs of the Newton loop: the finishing kernel stores its results in the pinned block and a sequence stamp behind them;
    // wait_results spins on the stamp (the host sees the stores a few microseconds before the runtime reports the kernel
    // complete: 1 650 waits per solve at L = 9) and falls back to hipStreamSynchronize after a millisecond or when polling is off.
    double next_seq() { return ++result_seq; }
    void wait_results(double seq);
    // fused line-search step (driver.cpp: trial_values): while set, the next evaluation on a selection level reads its point as
    // trial_x - trial_alpha * trial_dir instead of through the (not yet written) trial vector
    const double* trial_x = nullptr;
    const double* trial_dir = nullptr;
    double trial_alpha = 0.0;
    bool can_fuse_step(int level) const;
    // trial_fuse.want: the caller (trial_values) asks eval_f01_launch to run restriction, |g|^2 partials and the line-search step
    // (x, n, s -> xn, moved stamp) as ONE launch where the level allows it; .done reports that it happened
    struct TrialFuse { bool want = false, done = false; const double* x = nullptr; const double* n = nullptr; double s = 0.0;
                       double* xn = nullptr; int32_t* moved = nullptr; int32_t stamp = 0; } trial_fuse;
    // d_scal[lo .. lo + n) -> pin.d[lo .. lo + n), awaited (publish kernel + polled stamp)
    void read_scalars(int lo, int n);
    double result_seq = 0.0;
    std::mutex shared_mutex;               // prepare_all: state of the problem (not of one level) touched while levels are planned side by side
    bool prepared = false;
    void ensure_plan_dense(int level);
    double eval_f0(int level, const double* d_s, const double* d_zz, const double* d_cc);
    // kernels only: the value lands in d_scal[0]; the caller batches the read-back
    void eval_f0_launch(int level, const double* d_s, const double* d_zz, const double* d_cc);
    // d_part (sharded problems): receives this rank's partial gradient before the interface entries are summed
    void eval_f1(int level, const double* d_s, const double* d_zz, const double* d_cc, double* d_gout, double* d_part = nullptr);
    // one line-search trial: f0 (value in d_scal[0]) and f1 (gradient in d_gout) from one sweep over the elements
    // defer_f0_sum: leave the workgroup partials of f0 in d_partials for the caller's finishing launch (launch_trial_finish)
    void eval_f01_launch(int level, const double* d_s, const double* d_zz, const double* d_cc, double* d_gout, double* d_part = nullptr,
                         bool defer_f0_sum = false);
    // materialize = false (Newton loop): on direct levels only the shared entries are summed, H is not formed as a CSR
    // value array and the next factor(level, rhs) reads the slab (valid until the next eval_f2 of any level)
    // rhs (with materialize = false): the gradient the following factor(level, rhs) will carry; on levels with
    // condensed leaves (try_enable_condensed) the element kernel then writes the leaf fronts of the factorization itself
    void eval_f2(int level, const double* d_s, const double* d_zz, const double* d_cc, bool materialize = true,
                 const double* rhs = nullptr);
    bool try_enable_condensed(int level);
    int64_t hel_cap = 0;                   // doubles of the element-block slab region of d_hel; extras live behind it
    int hel_level = -1;                    // level whose blocks + shared sums d_hel currently holds (direct mode), or -1
    // returns MGBHIP_OK or MGBHIP_ERR_NOT_SPD; x = H^{-1} g on the device
    // rhs == nullptr: factor H for ordinary solves (trisolve).  rhs = g: factor the bordered matrix [H -g; -g' -1],
    // which performs the forward substitution of H x = g inside the factorization; trisolve_carried then needs the
    // backward sweep only (the Newton loop's one solve per factorization).
    void factor(int level, const double* rhs = nullptr);
    void trisolve(int level, const double* d_g, double* d_xout);
    void trisolve_carried(int level, double* d_xout_np1);      // output has room for m + 1 doubles
    // The reference's last resort when the symmetric factorizations fail (Julia's `Symmetric(H) \ g`: Cholesky -> LDL' -> LU,
    // src/utils.jl:142-145): dense LU with partial pivoting ON THE DEVICE (dense.hpp) for systems of at most DENSE_LU_MAX_M
    // unknowns whose H exists as a CSR value array.  d_xout = H^{-1} d_g.  Returns false when it does not apply (large or
    // slab-resident H, domain-decomposed problem) or the matrix is singular to working precision.
    bool lu_fallback(int level, const double* d_g, double* d_xout);
    mgbhip::DevBuf<double> d_lu;
    mgbhip::DevBuf<int32_t> d_lustat;
    int64_t lu_fallbacks = 0;              // how often the fallback produced the direction (diagnostics)
};
