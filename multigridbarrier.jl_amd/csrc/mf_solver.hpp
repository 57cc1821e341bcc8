// mf_solver.hpp -- device multifrontal sparse Cholesky: `n = solve(symmetric(H), g)`
// (reference: src/newton.jl:253, src/utils.jl:142-145; CUDA twin: cuDSS ANALYSIS once
// per pattern, FACTORIZATION + SOLVE per Newton iteration,
// ext/MultiGridBarrierCUDAExt/cudss_solver.jl:264-381).  Same life cycle here:
// analyze() once per sparsity pattern (host, cached by the level), factor() + solve()
// per Newton iteration on the device, all on the caller's stream.
#pragma once
#include <functional>

#include "common.hpp"
#include "mf_analysis.hpp"
#include "kernels.hpp"

namespace mgbhip {

struct FrontDev {
    int32_t k, m, nchild, a_cnt;
    int64_t F_off, idx_off, u_off, child_off, rel_off, a_off, acol_off;
    int64_t ug_off;        // large fronts: offset of the update-vector gather list (m + 1 pointers), -1 otherwise
    int32_t packed;        // leaf fronts (m <= 16) stored as a packed lower triangle: column c at c*m - c(c-1)/2, rows c..m-1
    int32_t pad_;
};

struct MfLaunch {          // one kernel launch: a contiguous range of fronts of one size class
    int32_t first, count;
    int32_t cls;           // LDS working size (0 = large-front multi-workgroup path)
    int32_t max_m, max_k;  // largest front / pivot block in the range
    int32_t max_child = 0; // most children of a front in the range
    bool tiny = false;     // leaf fronts with m <= 16: 16-lanes-per-front kernels
    bool wave = false;     // m <= 48 and only small children: one wave per front (mf_factor_wave), packed LDS triangle
    bool inv = false;      // large fronts on the inverse-based path (W_j = L_jj^{-1} in the arena, pivots in dvec)
    bool iface = false;    // the interface front of a domain-decomposed system, alone in its launch: assembled, summed
                           // over ranks (MfSolver::iface_reduce), then factored redundantly on every rank
};

class MfSolver {
   public:
    MfPlan plan;
    // coords (optional): n x dim row-major locations of the unknowns, an ordering hint (mf_analysis.hpp)
    void analyze(int64_t n, const int32_t* rowptr, const int32_t* colidx, hipStream_t st,
                 const double* coords = nullptr, int dim = 0, bool protect_peeled = false,
                 const int32_t* top = nullptr, int64_t ntop = 0);
    // Domain decomposition: called by factor() with the assembled interface front (device pointer, m*m doubles,
    // column-major, lower triangle meaningful) -- the caller sums it over ranks in place on the solver's stream.
    std::function<void(double*, int64_t)> iface_reduce;
    // factor the matrix whose CSR values (same pattern as analyze) live at d_values.
    // Asynchronous; the not-SPD flag is read back by status().
    // direct = true: d_values is the value space of set_direct_map (slab | shared | border) instead of the CSR array
    // condensed = true (needs direct and enable_condensed): the leaf fronts are already in the arena, written by the
    // element kernel of the same Newton iteration; the leaf level is skipped and no matrix entry is read
    void factor(const double* d_values, hipStream_t st, StageTimers* timers, bool direct = false, bool condensed = false);
    // Condensed leaves (kernels.hpp, launch_elem_f2_condense).  N elements of P nodes, ucol[e * P + i] = unknown of the
    // u component at node i of element e or -1, node P - 1 interior to the element, slack unknown of broken node q =
    // slack0 + q.  Checks that the plan's leaf level is exactly one front per element with the element's slacks and
    // interior node as pivots; builds the leaf descriptors and the border-only scatter lists.  nnz / tail_base as in
    // set_direct_map.  Returns false (and changes nothing) when the plan has another shape.
    bool enable_condensed(int64_t N, int P, const int32_t* ucol, int64_t slack0, int64_t nnz, int64_t tail_base, hipStream_t st);
    bool condensed_ready() const { return condensed_ok; }
    bool leaves_packed() const { return leaf_packed; }
    const LeafDesc* leaf_desc() const { return d_leaf_desc.p; }
    double* arena() { return d_arena.p; }
    int32_t* leaf_status() { return d_status.p + 1; }
    const int32_t* status_flags() const { return d_status.p; }      // [0] factorization, [1] leaf pivots (device)
    // The Newton loop reads both flags through a finishing kernel that also clears them (kernels.hpp, launch_dir_finish)
    // and tells the solver so: factor() / the condensing f2 then skip their hipMemsetAsync.
    int32_t* status_flags_rw() { return d_status.p; }
    void flags_cleared() { status_zero = true; leaf_zero = true; }
    bool leaf_flag_zero() const { return leaf_zero; }
    void leaf_flag_used() { leaf_zero = false; }
    // Second scatter list for the same plan: CSR position q -> value_map[q], border entry v -> tail_base + v.
    void set_direct_map(const int32_t* value_map, int64_t nnz, int64_t tail_base, hipStream_t st);
    bool has_direct_map() const { return d_a_src_direct.n > 0; }
    // Every system is factored BORDERED, [H c; c' gamma] (MfOptions::border): the value array passed to factor()
    // carries the border column in its tail, d_values[nnz + v] = c_v (v < n) and d_values[nnz + n] = gamma.
    //   c = 0, gamma = 1:  block diagonal; solve() is the ordinary forward + backward sweep of H x = b.
    //   c = -g, gamma = -1: the forward substitution of H x = g rides along the factorization as the last row of
    //                       every front; solve_border() is ONE backward sweep from x_n = 1 and returns H^{-1} g.
    // x = H^{-1} b (device vectors of length n; x may alias b); factors must come from a c = 0 border
    void solve(const double* d_b, double* d_x, hipStream_t st, StageTimers* timers);
    // x[0:n] = H^{-1} g for factors of the c = -g border; d_x_np1 has room for n + 1 doubles (x[n] = 1 on return)
    void solve_border(double* d_x_np1, hipStream_t st, StageTimers* timers);
    // Shape of one factorization + backward sweep as the device runs it (mgbhip_solver_chain): out[0] sequential
    // 32-column pivot blocks on the critical path of the large fronts (sum over their tree levels of ceil(max_k / 32)),
    // [1] tree levels on the large-front path, [2] kernel launches per factorization, [3] per backward sweep,
    // [4] doubles of the frontal arena, [5] factorization flops, [6] doubles the large fronts' trailing updates read + write
    // beyond one pass over the arena (the re-reads a tile-resident factorization would not make), [7] reserved
    void chain_stats(double* out8) const;
    int status(hipStream_t st);     // synchronises; MGBHIP_OK or MGBHIP_ERR_NOT_SPD
    // enqueue the copy of the flag only (pinned destination); interpret it after the caller's sync
    void status_async(int32_t* h_dst, hipStream_t st) const;
    static int status_from(const int32_t* flags2, bool with_leaves) {
        return (flags2[0] || (with_leaves && flags2[1])) ? MGBHIP_ERR_NOT_SPD : MGBHIP_OK;
    }
    bool factored_condensed = false;     // the current factors took their leaves from a condensing f2
    bool analyzed = false;
    // The inverse-based large-front kernels apply W = L_jj^{-1} where a substitution would run: as fast as a
    // GEMM, but only forward stable in cond(L_jj).  `robust` (set by the Newton loop when a direction fails its
    // sanity checks) makes factor() / solve() take the substitution-based kernels for every front instead.
    bool robust = false;
    bool has_inverse_path() const { return uses_inv; }
    // which representation the current factors are in (solve() must match the last factor())
    bool factored_inv = false;

   private:
    bool launch_big_assemble(const MfLaunch& L, dim3 grid, const double* d_values, const int32_t* a_src_p, hipStream_t st,
                             bool with_diag);     // returns true when block 0 of every front was factored by the launch
    void forward_pass(const double* d_b_np1, hipStream_t st, StageTimers* timers);
    void backward_pass(double* d_x_np1, hipStream_t st, StageTimers* timers);
    DevBuf<double> d_bx, d_xx, d_one;     // bordered right-hand side / solution of solve(), the constant 1
    DevBuf<FrontDev> d_fronts;
    DevBuf<int32_t> d_front_idx, d_children, d_rel, d_a_src, d_a_src_direct, d_a_dst, d_a_colptr;
    DevBuf<int64_t> d_ug_ptr, d_ug_src;   // per large front: for every local index the children's update-vector entries, in child order
    DevBuf<double> d_arena, d_uvec, d_y, d_tbig, d_tsol, d_dscr, d_dvec;
    DevBuf<double> d_ifpack;              // packed lower triangle of the interface front (domain decomposition)
    DevBuf<int32_t> d_status;
    // condensed leaves
    bool condensed_ok = false;
    bool leaf_packed = false;            // the m <= 16 leaf fronts are stored as packed triangles (FrontDev::packed)
    std::vector<FrontDev> h_fronts;
    std::vector<int32_t> h_a_dst;         // a_dst as uploaded (packed-triangle remaps applied)
    DevBuf<FrontDev> d_fronts_c;
    DevBuf<int32_t> d_a_src_c, d_a_dst_c, d_a_colptr_c;
    DevBuf<LeafDesc> d_leaf_desc;
    const FrontDev* cur_fr = nullptr;     // arrays of the factorization in progress
    const int32_t* cur_adst = nullptr;
    const int32_t* cur_acol = nullptr;
    std::vector<std::vector<MfLaunch>> level_launches;   // per level, leaves first (factorization: one per LDS class)
    std::vector<std::vector<MfLaunch>> level_solves;     // triangular solves: all LDS-class fronts of a level in one launch
    int32_t lds_cap = 88;           // largest m factored out of LDS
    bool uses_inv = false;
    bool y_border_one = false;      // d_y[n] holds the 1 the border sweep starts from
    bool y_zero = false;            // d_y[0:n) is zero (the backward sweeps only read it; a generic forward sweep overwrites it)
    bool status_zero = false, leaf_zero = false;    // d_status[0] / [1] are known to be zero on the stream
};

}  // namespace mgbhip
