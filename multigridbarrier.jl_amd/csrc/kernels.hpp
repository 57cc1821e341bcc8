// kernels.hpp -- launch wrappers of the evaluate/assemble kernels (kernels.hip).
#pragma once
#include "common.hpp"
#include "cone.hpp"

namespace mgbhip {

enum ElemMode { MODE_F0 = 0, MODE_F1 = 1, MODE_F2 = 2, MODE_NODE_F = 3, MODE_NODE_SLACK = 4,
                MODE_F01 = 5 };   // MODE_F01: value and gradient of one line-search trial in ONE pass over the operators

// Where the leaf front of an element lives and how its boundary is ordered (built by MfSolver::enable_condensed).
struct LeafDesc {
    int64_t F_off;       // offset of the frontal matrix in the arena (column-major m x m)
    int32_t interior;    // column (unknown id) of the element-interior u node
    uint32_t packed;     // bits 0-3: m; bits 4+4i .. 7+4i: position of element node i in the front's index list, 15 = no unknown
};

struct ElemParams {
    int32_t p, nu, nD, nstage;
    int32_t ymask;                           // bit k set: row k of y enters some barrier term (else d/dy_k = 0)
    int64_t N, n;
    const double* ops[MGBHIP_MAX_OPS];       // device operator arrays (nullptr = identity)
    const double* stage_ptr[MGBHIP_MAX_OPS]; // operators staged through LDS (distinct, non-identity)
    int32_t D_state[MGBHIP_MAX_ND];
    int32_t D_op[MGBHIP_MAX_ND];             // index into ops
    int32_t D_stage[MGBHIP_MAX_ND];          // -1 identity, >= 0 slot in stage_ptr, -2 read from HBM
    const double* w;
    const double* c;                         // n x nD (may be nullptr for the node maps)
    const double* z0;                        // nu*n: the fine broken-basis iterate z0 + R*s (already prolonged)
    const double* bw;                        // barrier weights or nullptr
    double invn;
    ConeDev cone;
    double* out_partial;                     // f0: one partial per workgroup
    double* out_ret;                         // f1: nu*n vector sum_k D_k' Y_k
    double* out_hel;                         // f2: element Hessian blocks
    double* out_F;                           // node maps
    double* out_Dz;
    int32_t diag_mask;                       // bit blk set: element block blk is diagonal and stored compactly (p per element)
    int64_t blk_off[MGBHIP_MAX_NU * (MGBHIP_MAX_NU + 1) / 2];   // slab offset of every element block
    double* dn_Dz;                           // dense path (p > 64, N = 1): n x nD workspace for D*z
    double* dn_Y;                            // dense path: n x nD (f1) / n x nD(nD+1)/2 (f2) node weights
    // condensing f2 (launch_elem_f2_condense): the element kernel eliminates the element-local unknowns itself
    const LeafDesc* leaf_desc;               // one per element
    double* leaf_arena;                      // the solver's frontal arena
    const double* leaf_g;                    // gradient in level coefficients (the border column is -g)
    int64_t leaf_slack0;                     // column of the slack unknown of broken node 0 (slack of node i = leaf_slack0 + i)
    int32_t* leaf_status;                    // |= 1 on a zero / non-finite pivot
    int32_t leaf_packed;                     // leaf fronts are packed lower triangles (FrontDev::packed)
    // selection levels (every row of R has at most one entry, equal to 1): z = z0 + R s is formed inside the element
    // kernels -- zsel[i] is the column of row i or -1, zs the level's unknown vector -- instead of by a prolongation launch
    const int32_t* zsel;
    const double* zs;
    double* zout;                            // with zsel: the kernel also stores z0 + R s (the next evaluation at this point reads it)
    // line-search trial on a selection level: s is the trial point x - alpha n formed on the fly, zs = x, zx = n (the step kernel
    // that materialises it runs BEHIND this evaluation instead of in front of it, off the host round trip's critical path)
    const double* zx;
    double zalpha;
};

// Fine-level Newton systems: H of the default problem couples the p broken slack unknowns of an element (diagonal
// H_ss because D_s = id) and its interior u unknown (the bubble) to the element's own nodes only -- the leaf fronts of
// the elimination tree (mf_analysis.hpp, simplicial peeling) are one per element.  launch_elem_f2_condense computes the
// element blocks like MODE_F2 and performs that leaf's partial factorization in place (static condensation of slacks
// and bubble, with the border row of the bordered system [H -g; -g' -1]), writing the leaf front -- L panel, pivots,
// update block -- straight into the solver's arena: the element blocks never reach HBM, the leaf level of the
// factorization and the shared-entry sums of the assembly disappear.  Returns false when no specialisation applies.
bool launch_elem_f2_condense(const ElemParams& P, hipStream_t st);

// element Hessian slab layout: block-major [block][element][p*p]; blocks (a,b), a <= b, in
// row-major order of the upper block triangle, each p x p column-major.
inline int hel_blocks(int nu) { return nu * (nu + 1) / 2; }
inline int hel_block_index(int a, int b, int nu) { return a * nu - a * (a - 1) / 2 + (b - a); }
// Offsets of the blocks in the slab: full blocks hold p*p doubles per element, blocks flagged in
// diag_mask (both states carry identity operators only) hold their p diagonal entries.
inline int64_t hel_layout(int nu, int64_t N, int p, int diag_mask, int64_t* off) {
    int64_t total = 0;
    for (int blk = 0; blk < hel_blocks(nu); ++blk) {
        off[blk] = total;
        total += ((diag_mask >> blk) & 1) ? N * p : N * (int64_t)p * p;
    }
    return total;
}

int elem_group(int p);                                   // lanes per element (power of two >= p)
int64_t elem_grid(int p, int64_t N);                     // workgroups
size_t elem_lds_bytes(const ElemParams& P, int mode);
void launch_elem(const ElemParams& P, int mode, hipStream_t st);

// deterministic two-stage reductions into a device scalar block
void launch_reduce_partials(const double* partials, int64_t count, double* out, hipStream_t st);
// stats[0] = sum v^2, stats[1] = number of non-finite entries (as double)
// mask (optional, domain decomposition): weights of the sums, 1 on the entries this rank owns and 0 elsewhere
// ints (optional): nints device flags copied behind the sums as doubles (stats[2 ..] / stats3[3 ..]): one read-back for all
void launch_vec_stats(const double* v, int64_t n, double* scratch, double* stats, hipStream_t st, const double* mask = nullptr,
                      const int32_t* ints = nullptr, int nints = 0);
// stats3 = [sum v*v, count of non-finite v, g.v] in one pass (values identical to launch_vec_stats + launch_dot)
void launch_dir_stats(const double* v, const double* g, int64_t n, double* scratch, double* stats3, hipStream_t st,
                      const double* mask = nullptr, const int32_t* ints = nullptr, int nints = 0);
void launch_dot(const double* a, const double* b, int64_t n, double* scratch, double* out, hipStream_t st, const double* mask = nullptr);
// The two read-backs of a Newton iteration, each ONE finishing launch that also writes the pinned host block `host`
// (device-visible pointer of a hipHostMalloc block; same slot indices as `scal`) -- no copy launch, no memset launch:
//   direction: scal[2] = sum v^2, [3] = non-finite count of v, [4] = g.v, [5], [6] = status2[0..1] (read, then cleared)
//   trial:     scal[0] = sum of f0_partials (nullptr: scal[0] is already final), [2] = sum g^2, [3] = non-finite count, [4] = *moved
// restriction + |g|^2 partials (+ the line-search step when xn != nullptr) in one launch; scratch gets the partial sums in the layout
// launch_trial_finish(partials_ready = true) reads
void launch_restrict_trial(int64_t rows, const int32_t* ptr, const int32_t* col, const double* val, const double* ret, double* g,
                            double* scratch, const double* x, const double* nn, double s, double* xn, int32_t* moved, int32_t stamp,
                            hipStream_t st);
void launch_publish(const double* src, int n, double* host_block, int host_lo, double seq, hipStream_t st);
void launch_dir_finish(const double* v, const double* g, int64_t n, double* scratch, double* scal, int32_t* status2, double* host,
                       hipStream_t st, const double* mask = nullptr, double seq = 0.0);
void launch_trial_finish(const double* g, int64_t n, double* scratch, const double* f0_partials, int64_t f0_count, double* scal,
                         int32_t* moved, double* host, hipStream_t st, const double* mask = nullptr, double seq = 0.0,
                         bool partials_ready = false);
void launch_index_gather(const double* v, const int32_t* idx, int64_t cnt, double* out, hipStream_t st);     // out[i] = v[idx[i]]
void launch_index_scatter(const double* in, const int32_t* idx, int64_t cnt, double* v, hipStream_t st);    // v[idx[i]] = in[i]
int64_t reduce_scratch_doubles(int64_t n);

// y[i] = sum_j A[i,j] x[j]  (CSR, deterministic; wave-per-row when rows are long)
void launch_csr_matvec(int64_t rows, const int32_t* ptr, const int32_t* col, const double* val,
                       const double* x, double* y, bool add, bool long_rows, hipStream_t st);
// rows of more than ~8K entries: (row, chunk) workgroups + fixed-order sum; scratch holds rows*nchunk doubles
int csr_chunks(int64_t max_row_len);
void launch_csr_matvec_chunked(int64_t rows, const int32_t* ptr, const int32_t* col, const double* val,
                               const double* x, double* y, double* scratch, int nchunk, hipStream_t st);
void launch_prolong(int64_t rows, const int32_t* ptr, const int32_t* col, const double* val, const double* s,
                    const double* z0, double* zfull, hipStream_t st);
// xn = x - s*n ; flag[0] = stamp if any(xn != x)  (a fresh stamp per launch: the flag never needs clearing)
void launch_step(const double* x, const double* n, double s, double* xn, int64_t len, int32_t* moved, int32_t stamp,
                 hipStream_t st);
void launch_scale_copy(const double* src, double alpha, double* dst, int64_t len, hipStream_t st);
void launch_border_tail(const double* g, double* tail, int64_t m, hipStream_t st);   // tail = [-g; -1]
void launch_axpy(double alpha, const double* x, double* y, int64_t len, hipStream_t st);
void launch_fill(double value, double* y, int64_t len, hipStream_t st);       // y[:] = value

// selection levels: H[q] = sum_{t in contributions(q)} slab[cidx[t]]
// sums of the structural nonzeros with other than one contribution (direct-value levels), in list order
void launch_gather_shared(int64_t nshared, const int32_t* sh_q, const int32_t* cptr, const int32_t* cidx, const double* slab,
                          double* out, hipStream_t st);
// qmap / nq (optional): assemble only the listed positions (the Newton loop: the upper triangle, all the solver reads)
void launch_gather_assemble(int64_t nnz, const int32_t* cptr, const int32_t* cidx, const double* slab,
                            double* Hval, bool long_lists, hipStream_t st, int32_t chunk = 0, int32_t nchunk = 0,
                            double* part = nullptr, const int32_t* qmap = nullptr, int64_t nq = 0);     // nchunk > 1: two-stage sums of very long lists

// general (coarse) levels: slab_e = [panel_0 .. panel_{nu-1}]' * Hel_e * [panel_0 .. panel_{nu-1}]
// (c_tot x c_tot, column-major, at eoff[e]); the structural nonzeros then gather from the slab.
struct PanelParams {
    int32_t p, nu;
    int64_t N;
    const int32_t* ecol_ptr;      // [N*nu + 1] offsets of the per-(element, state) column lists
    const double* panels;         // per (element, state): p x c panel, column-major, at p * ecol_ptr[...]
    const int32_t* eoff;          // [N + 1] slab offsets
    const double* hel;
    double* slab;
    int32_t cmax;                 // largest per-(element, state) column count
    int32_t upper_only;           // project only the entries (i <= j) of the element's block: what an upper-triangle gather reads
    const int32_t* spos;          // [slab doubles] or nullptr: where entry (element offset + i + ct j) goes -- its place in the
                                  // contribution list of its Hessian entry, so that the gather reads contiguous runs
};
// one 16 x 16 output tile per matrix-core instruction pair: T = Hel P and P' T on v_mfma_f64_16x16x4 (one wave or one
// workgroup per element); false if the element's staging does not fit (the loop kernels above then run)
bool launch_panel_project_mfma(const PanelParams& P, hipStream_t st);
// spos[cidx[t]] = t for t < total (cidx is a permutation of the slab positions of a projected level)
void launch_invert_lists(const int32_t* cidx, int64_t total, int32_t* spos, hipStream_t st);
void launch_panel_project(const PanelParams& P, hipStream_t st);
// small coarse levels with wide supports: element streams x chunks of the packed upper triangle of H
// accumulated in LDS, then a fixed-order sum over the streams.  H is the dense m x m array (both
// triangles written); partial holds nstream * m(m+1)/2 doubles; chunk * nsplit >= m(m+1)/2.
constexpr size_t PANEL_ACC_LDS_MAX = 144 * 1024;
size_t panel_accumulate_lds(int p, int nu, int ctmax);     // slab variant: 4 waves x one element's staging
size_t panel_stage_doubles(int p, int nu, int ctmax);      // one element's staging in doubles
bool panel_accumulate_fits(int p, int nu, int ctmax);       // register-staging limits of the accumulate kernel
void launch_panel_accumulate(const PanelParams& P, const int32_t* ecols, int32_t m, int32_t nstream, int32_t nsplit,
                             int32_t chunk, int32_t ctmax, double* partial, double* H, hipStream_t st);
// same projection as launch_panel_project from LDS-staged panels
void launch_panel_project_staged(const PanelParams& P, int32_t ctmax, hipStream_t st);

}  // namespace mgbhip
