// mf_analysis.hpp -- symbolic analysis for the multifrontal sparse Cholesky that
// replaces the reference's `solve(symmetric(H), g)` (reference: src/newton.jl:253,
// src/utils.jl:142-145 -> Julia stdlib CHOLMOD; ext/MultiGridBarrierCUDAExt/cudss_solver.jl
// on CUDA).  Pure host C++ (no HIP), so the ordering / tree / index maps can be unit
// tested on the CPU; the numeric phase runs on the device (mf_numeric.hip).
//
// Pipeline
//   1. simplicial peeling: repeatedly order nodes whose remaining neighbourhood is a
//      clique (zero fill).  On the fine level this removes the broken slack DoFs
//      (H_ss is diagonal because D_s = id) and then the element bubbles, i.e. the
//      classical FEM static condensation, discovered from the graph alone;
//   2. nested dissection of the rest: geometric (coordinate-median cuts, separator = the
//      boundary layer of one side) when the caller knows where the unknowns live, else BFS
//      level-set bisection with separator trimming; leaves of <= leaf_size nodes;
//   3. supernodal symbolic factorization on that partition (exact row structures),
//      exact-fit amalgamation of a child into its parent, relative index maps,
//      level schedule (leaves first).
#pragma once
#include <cstdint>
#include <vector>

namespace mgbhip {

struct Front {
    int32_t k = 0;            // pivots
    int32_t m = 0;            // pivots + boundary
    int64_t idx_off = 0;      // into MfPlan::front_idx (m entries: global node ids)
    int64_t F_off = 0;        // into the frontal-matrix arena (m*m doubles, column-major, ld = m)
    int64_t u_off = 0;        // into the update-vector arena (m - k doubles)
    int32_t parent = -1;
    int32_t level = 0;
    int64_t child_off = 0;    // into MfPlan::children
    int32_t nchild = 0;
    int64_t rel_off = 0;      // into MfPlan::rel: (m-k) positions of this front's boundary in its parent
    int64_t a_off = 0;        // into MfPlan::a_src / a_dst
    int32_t a_cnt = 0;
    int64_t acol_off = 0;     // into MfPlan::a_colptr (k + 1 entries: A entries of pivot column c are [ptr[c], ptr[c+1]))
};

struct MfPlan {
    int64_t n = 0;                      // unknowns of H (the border, if any, is unknown n)
    bool border = false;
    std::vector<Front> fronts;          // sorted by (level, size class)
    std::vector<int32_t> front_idx;     // concatenated index lists (global node ids)
    std::vector<int32_t> children;      // concatenated child front ids
    std::vector<int32_t> rel;           // concatenated relative indices
    std::vector<int32_t> a_src;         // CSR value index of H feeding a_dst
    std::vector<int32_t> a_dst;         // position row + col*m inside the front
    std::vector<int32_t> a_colptr;      // per front, per pivot column: offsets into its A entries (relative to a_off)
    std::vector<int32_t> level_ptr;     // fronts [level_ptr[l], level_ptr[l+1]) are level l
    int64_t arena_doubles = 0;          // sum m*m
    int64_t uvec_doubles = 0;           // sum (m-k)
    int32_t max_m = 0;
    int64_t factor_flops = 0;
    // statistics
    int64_t peeled = 0;
    int32_t peel_rounds = 0;
    int32_t iface_front = -1;           // MfOptions::top: the front holding the interface unknowns
};

struct MfOptions {
    int32_t leaf_size = 24;       // measured best on MI355X (fem2d_P2 L=7..9); the optimum is flat between 16 and 32
    int32_t max_peel_rounds = 4;
    int32_t peel_max_degree = 48;
    double sep_weight = 1.5;      // separator-size penalty in the bisection score
    int32_t merge_max_m = 0;      // relaxed amalgamation: merge a child into its parent while m stays <= this
    int32_t exact_merge_max_m = 128;  // exact-fit amalgamation only while the merged front stays this small
    // Bordered system [H c; c' gamma]: one extra unknown (id n), adjacent to every other and eliminated last.
    // It is appended to the boundary of every front and gets a 1 x 1 root front of its own.  With c = -g the
    // factorization carries the forward substitution of H x = g along as an extra row of every front, and a
    // backward sweep from x_n = 1 returns x = H^{-1} g (MfSolver::solve_border); with c = 0, gamma = 1 the
    // bordered system is block diagonal and ordinary solves are unchanged.  Values of the border column are read
    // from the tail of the value array: entry (v, n) at nnz + v, entry (n, n) at nnz + n.
    bool border = false;
    // Keep fronts made of peeled unknowns apart from the dissection fronts above them (no exact-fit amalgamation
    // across that line): on fine levels the peeled fronts are then exactly the per-element static-condensation leaves
    // the condensing element kernel writes (MfSolver::enable_condensed).
    bool protect_peeled = false;
    // Domain decomposition (one process per GPU): `top` lists the interface unknowns of this
    // rank's local system, in an order common to all ranks.  They are kept out of peeling and dissection and form ONE
    // final front (k = |top|, boundary = the border), never amalgamated with anything: its assembled frontal matrix is
    // the rank's contribution to the interface Schur complement, summed over ranks before it is factored.
    const int32_t* top = nullptr;
    int64_t ntop = 0;
};

// Symmetric pattern in CSR (both triangles present, diagonal optional).  Values are not
// needed: the plan records, for every structural pair {v,u}, the CSR position of the
// entry in the upper triangle (row <= col), which is what `symmetric(H)` reads.
// coords (optional): n x dim row-major locations of the unknowns; when given, the dissection cuts
// the longest axis of each subset's bounding box at the median (straight separators on meshes).
void mf_analyze(int64_t n, const int32_t* rowptr, const int32_t* colidx, const MfOptions& opt,
                MfPlan& plan, const double* coords = nullptr, int dim = 0);

}  // namespace mgbhip
