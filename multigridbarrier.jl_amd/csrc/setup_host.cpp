// setup_host.cpp -- libmgbsetup.so: host-side helpers of the Python setup layer (row f4, time to first solution).
// Plain C++ (g++, no HIP): the sequential loops of the AMG hierarchy construction that NumPy cannot vectorise (Ruge-Stueben
// splitting, row sorts) and the ladder composition as a chain of sparse products that runs without the interpreter lock.
// Each has a pure-Python twin in the package that tests/test_setup.py compares bit for bit; the package falls back to
// the twin when this library has not been built (the setup layer is host code either way -- the Newton path has no fallback).
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

extern "C" {

// Ruge-Stueben first-pass C/F splitting with the bucket ordering of AlgebraicMultigrid.jl's RS_CF_splitting (what the
// reference's `amg_ruge_stuben` runs, src/amg_prolongators.jl:16-18); statement and citations: amg_prolongators.py:
// _rs_cf_splitting, of which this is the line-by-line twin.  S row i (Sp, Sj): the nodes i strongly depends on, sorted;
// T = S' (Tp, Tj).  is_c[i] = 1 for C points.
int mgbsetup_rs_cf_splitting(int64_t n, const int32_t* Sp, const int32_t* Sj, const int32_t* Tp, const int32_t* Tj,
                             int32_t diag_quirk, uint8_t* is_c) {
    if (n < 0 || (n > 0 && (!Sp || !Tp || !is_c))) return 1;
    enum : uint8_t { F_NODE = 0, C_NODE = 1, U_NODE = 2 };
    std::vector<int64_t> lam((size_t)n), interval_count((size_t)n + 2, 0), interval_ptr((size_t)n + 2, 0);
    std::vector<int64_t> index_to_node((size_t)n), node_to_index((size_t)n);
    std::vector<uint8_t> split((size_t)n, U_NODE);
    for (int64_t i = 0; i < n; ++i) {
        lam[i] = Tp[i + 1] - Tp[i];
        interval_count[lam[i]]++;
    }
    int64_t cs = 0;
    for (int64_t l = 0; l <= n; ++l) {
        interval_ptr[l] = cs;
        cs += interval_count[l];
        interval_count[l] = 0;
    }
    for (int64_t i = 0; i < n; ++i) {
        const int64_t l = lam[i], idx = interval_ptr[l] + interval_count[l];
        index_to_node[idx] = i;
        node_to_index[i] = idx;
        interval_count[l]++;
    }
    for (int64_t i = 0; i < n; ++i) {
        if (lam[i] == 0) split[i] = F_NODE;
        else if (diag_quirk && lam[i] == 1 && (Sp[i] == Sp[i + 1] || Sj[Sp[i]] > i)) split[i] = F_NODE;
    }
    for (int64_t top = n - 1; top >= 0; --top) {
        const int64_t i = index_to_node[top];
        interval_count[lam[i]]--;
        if (split[i] == F_NODE) continue;
        split[i] = C_NODE;
        for (int32_t jj = Tp[i]; jj < Tp[i + 1]; ++jj) {            // nodes that depend on the new C point
            const int64_t j = Tj[jj];
            if (split[j] != U_NODE) continue;
            split[j] = F_NODE;
            for (int32_t kk = Sp[j]; kk < Sp[j + 1]; ++kk) {        // what the new F point depends on
                const int64_t k = Sj[kk];
                if (split[k] != U_NODE || lam[k] >= n - 1) continue;
                const int64_t lk = lam[k], old = node_to_index[k], nw = interval_ptr[lk] + interval_count[lk] - 1;
                const int64_t a = index_to_node[old], b = index_to_node[nw];
                node_to_index[a] = nw; node_to_index[b] = old;
                index_to_node[old] = b; index_to_node[nw] = a;
                interval_count[lk]--;
                interval_count[lk + 1]++;
                interval_ptr[lk + 1] = nw;
                lam[k] = lk + 1;
            }
        }
        for (int32_t jj = Sp[i]; jj < Sp[i + 1]; ++jj) {            // what the new C point depends on
            const int64_t j = Sj[jj];
            if (split[j] != U_NODE || lam[j] == 0) continue;
            const int64_t lj = lam[j], old = node_to_index[j], nw = interval_ptr[lj];
            const int64_t a = index_to_node[old], b = index_to_node[nw];
            node_to_index[a] = nw; node_to_index[b] = old;
            index_to_node[old] = b; index_to_node[nw] = a;
            interval_count[lj]--;
            interval_count[lj - 1]++;
            interval_ptr[lj]++;
            interval_ptr[lj - 1] = interval_ptr[lj] - interval_count[lj - 1];
            lam[j] = lj - 1;
        }
    }
    for (int64_t i = 0; i < n; ++i) is_c[i] = split[i] == C_NODE ? 1 : 0;
    return 0;
}

// Sort the column indices of every CSR row (values follow), in place: what scipy's `sort_indices` does, without its
// per-row temporary of pairs -- the composed prolongators have 3-4 entries per row and ten of them are sorted per ladder.
// Rows are expected duplicate-free (a stable order among equal columns is kept anyway).
int mgbsetup_csr_sort_rows(int64_t nrows, const int32_t* ptr, int32_t* idx, double* val) {
    if (nrows < 0 || (nrows > 0 && !ptr)) return 1;
    std::vector<std::pair<int32_t, double>> tmp;
    for (int64_t i = 0; i < nrows; ++i) {
        const int64_t lo = ptr[i], hi = ptr[i + 1], len = hi - lo;
        if (len < 2) continue;
        if (len <= 16) {                                    // insertion sort (stable)
            for (int64_t a = lo + 1; a < hi; ++a) {
                const int32_t c = idx[a];
                const double v = val[a];
                int64_t b = a;
                while (b > lo && idx[b - 1] > c) { idx[b] = idx[b - 1]; val[b] = val[b - 1]; --b; }
                idx[b] = c; val[b] = v;
            }
        } else {
            tmp.resize((size_t)len);
            for (int64_t a = 0; a < len; ++a) tmp[(size_t)a] = {idx[lo + a], val[lo + a]};
            std::stable_sort(tmp.begin(), tmp.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
            for (int64_t a = 0; a < len; ++a) { idx[lo + a] = tmp[(size_t)a].first; val[lo + a] = tmp[(size_t)a].second; }
        }
    }
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Ladder composition (reference: src/multigrid.jl:192-204, `refine_fine_prod[l] = refine_fine_prod[l+1] * refine[l]`).
// A chain handle holds the running product C (CSR, rows in the storage order scipy's `csr_matmat` produces: the reverse
// of the order in which a row's columns are first touched -- the NEXT product sums in that order, so it is part of the
// result) and multiplies it by one factor per step with exactly scipy's kernel: same loop, same accumulator array, same
// dropped exact zeros, no fused multiply-add (built with -ffp-contract=off).  `emit_sorted` writes the current product with
// sorted rows into caller-owned arrays (what `_as_op` hands to the problem).  One chain per ladder, so the ladders of a
// hierarchy (full / dirichlet) run on two Python threads: ctypes releases the GIL, scipy's sparsetools do not.
namespace {
struct Chain {
    int64_t rows = 0, cols = 0;
    std::vector<int32_t> ptr, idx;
    std::vector<double> val;
    // Rows of the first operand that are identical entry for entry (the broken P2 nodes that share a mesh node carry the same
    // row of the bridge) stay identical in every product of the chain: the products are formed on the DISTINCT rows only, in
    // order of first occurrence, and written out through this map -- the same bits for 43 % of the work at L = 9.
    std::shared_ptr<std::vector<int32_t>> rowmap;      // empty: the stored rows are the rows
    int64_t full_rows = 0;
};

// Distinct rows by one pass over an open-addressing table keyed by a hash of the row's entries; a hit is verified entry for
// entry (indices and the values' bits) before the row takes the earlier one as its representative.  rep_of_row[i] = index of
// row i's representative among `reps` (the distinct rows in order of first occurrence).
static void distinct_rows(const Chain& A, std::vector<int32_t>& rep_of_row, std::vector<int32_t>& reps) {
    const int64_t n = A.rows;
    size_t cap = 1;
    while (cap < (size_t)(2 * n + 16)) cap <<= 1;
    std::vector<int32_t> table(cap, -1);                // row index of the representative stored in a slot
    auto same = [&](int32_t a, int32_t b) {
        const int32_t la = A.ptr[a + 1] - A.ptr[a];
        if (la != A.ptr[b + 1] - A.ptr[b]) return false;
        if (la == 0) return true;
        return std::memcmp(&A.idx[A.ptr[a]], &A.idx[A.ptr[b]], sizeof(int32_t) * (size_t)la) == 0 &&
               std::memcmp(&A.val[A.ptr[a]], &A.val[A.ptr[b]], sizeof(double) * (size_t)la) == 0;
    };
    rep_of_row.assign((size_t)n, -1);
    reps.clear();
    for (int64_t i = 0; i < n; ++i) {
        uint64_t h = 1469598103934665603ull;
        for (int32_t q = A.ptr[i]; q < A.ptr[i + 1]; ++q) {
            uint64_t bits;
            std::memcpy(&bits, &A.val[q], sizeof(bits));
            h ^= (uint64_t)(uint32_t)A.idx[q] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
            h ^= bits + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
        }
        size_t slot = (size_t)(h * 0x9e3779b97f4a7c15ull) & (cap - 1);
        while (true) {
            const int32_t r = table[slot];
            if (r < 0) {                                // new distinct row
                table[slot] = (int32_t)i;
                rep_of_row[(size_t)i] = (int32_t)reps.size();
                reps.push_back((int32_t)i);
                break;
            }
            if (same(r, (int32_t)i)) { rep_of_row[(size_t)i] = rep_of_row[(size_t)r]; break; }
            slot = (slot + 1) & (cap - 1);
        }
    }
}
}  // namespace

// stored entries of the product as the caller sees it (all rows, not only the distinct ones)
static int64_t full_nnz(const Chain& c) {
    if (!c.rowmap) return (int64_t)c.idx.size();
    int64_t n = 0;
    for (int32_t r : *c.rowmap) n += c.ptr[(size_t)r + 1] - c.ptr[(size_t)r];
    return n;
}

void* mgbsetup_chain_create(int64_t rows, int64_t cols, const int32_t* ptr, const int32_t* idx, const double* val) {
    if (rows < 0 || cols < 0 || !ptr) return nullptr;
    Chain* c = new Chain();
    c->rows = rows; c->cols = cols;
    c->full_rows = rows;
    c->ptr.assign(ptr, ptr + rows + 1);
    const int64_t nnz = ptr[rows];
    if (nnz > 0) { c->idx.assign(idx, idx + nnz); c->val.assign(val, val + nnz); }
    const char* env_min = std::getenv("MGB_SETUP_DISTINCT_MIN_ROWS");       // tests lower it to reach this path on small meshes
    if (rows >= (env_min ? std::atoll(env_min) : 100000ll)) {   // large first operand: keep its distinct rows only, if that is a real saving
        std::vector<int32_t> rep_of_row, reps;
        distinct_rows(*c, rep_of_row, reps);
        if ((int64_t)reps.size() * 4 <= rows * 3) {
            Chain u;
            u.rows = (int64_t)reps.size(); u.cols = cols; u.full_rows = rows;
            u.ptr.assign(reps.size() + 1, 0);
            for (size_t t = 0; t < reps.size(); ++t) {
                const int32_t r = reps[t];
                u.idx.insert(u.idx.end(), c->idx.begin() + c->ptr[r], c->idx.begin() + c->ptr[r + 1]);
                u.val.insert(u.val.end(), c->val.begin() + c->ptr[r], c->val.begin() + c->ptr[r + 1]);
                u.ptr[t + 1] = (int32_t)u.idx.size();
            }
            u.rowmap = std::make_shared<std::vector<int32_t>>(std::move(rep_of_row));
            *c = std::move(u);
        }
    }
    return c;
}

void mgbsetup_chain_destroy(void* h) { delete static_cast<Chain*>(h); }

// A * B with scipy's kernel (see above); false if the product exceeds 32-bit indexing
static bool chain_product(const Chain& A, int64_t bcols, const int32_t* Bp, const int32_t* Bj, const double* Bx, Chain& out) {
    std::vector<int32_t> next((size_t)bcols, -1);
    std::vector<double> sums((size_t)bcols, 0.0);
    out.rows = A.rows; out.cols = bcols;
    out.rowmap = A.rowmap; out.full_rows = A.full_rows;
    out.ptr.assign((size_t)A.rows + 1, 0);
    out.idx.clear(); out.val.clear();
    {   // candidates per row bound the product: reserve once (untouched pages cost nothing), never reallocate
        size_t ub = 0;
        for (size_t q = 0; q < A.idx.size(); ++q) ub += (size_t)(Bp[A.idx[q] + 1] - Bp[A.idx[q]]);
        out.idx.reserve(ub);
        out.val.reserve(ub);
    }
    for (int64_t i = 0; i < A.rows; ++i) {
        int32_t head = -2, length = 0;
        for (int32_t jj = A.ptr[i]; jj < A.ptr[i + 1]; ++jj) {
            const int32_t j = A.idx[jj];
            const double v = A.val[jj];
            for (int32_t kk = Bp[j]; kk < Bp[j + 1]; ++kk) {
                const int32_t k = Bj[kk];
                sums[k] += v * Bx[kk];
                if (next[k] == -1) { next[k] = head; head = k; ++length; }
            }
        }
        for (int32_t t = 0; t < length; ++t) {
            if (sums[head] != 0) { out.idx.push_back(head); out.val.push_back(sums[head]); }
            const int32_t tmp = head;
            head = next[head];
            next[tmp] = -1;
            sums[tmp] = 0;
        }
        if (out.idx.size() >= (size_t)INT32_MAX) return false;
        out.ptr[(size_t)i + 1] = (int32_t)out.idx.size();
    }
    return true;
}

// C <- C * B (B: brows x bcols CSR, brows == cols of C).  Returns the number of stored entries of the product, -1 on a shape
// mismatch, -2 if the product exceeds 32-bit indexing.
int64_t mgbsetup_chain_multiply(void* h, int64_t brows, int64_t bcols, const int32_t* Bp, const int32_t* Bj, const double* Bx) {
    Chain* c = static_cast<Chain*>(h);
    if (!c || brows != c->cols || bcols < 0 || !Bp) return -1;
    Chain out;
    if (!chain_product(*c, bcols, Bp, Bj, Bx, out)) return -2;
    *c = std::move(out);
    return full_nnz(*c);
}

int64_t mgbsetup_chain_nnz(void* h) { return h ? full_nnz(*static_cast<Chain*>(h)) : -1; }

// The current product with sorted rows into ptr[rows + 1], idx[nnz], val[nnz] (caller-owned).
int mgbsetup_chain_emit_sorted(void* h, int32_t* ptr, int32_t* idx, double* val) {
    Chain* c = static_cast<Chain*>(h);
    if (!c || !ptr) return 1;
    if (c->rowmap) {                                    // sort the distinct rows once, then write every row through the map
        Chain sorted = *c;
        if (mgbsetup_csr_sort_rows(sorted.rows, sorted.ptr.data(), sorted.idx.data(), sorted.val.data()) != 0) return 1;
        const std::vector<int32_t>& map = *c->rowmap;
        int64_t w = 0;
        ptr[0] = 0;
        for (int64_t i = 0; i < c->full_rows; ++i) {
            const int32_t r = map[(size_t)i];
            const int32_t lo = sorted.ptr[r], len = sorted.ptr[r + 1] - lo;
            std::copy(sorted.idx.begin() + lo, sorted.idx.begin() + lo + len, idx + w);
            std::copy(sorted.val.begin() + lo, sorted.val.begin() + lo + len, val + w);
            w += len;
            ptr[i + 1] = (int32_t)w;
        }
        return 0;
    }
    std::copy(c->ptr.begin(), c->ptr.end(), ptr);
    if (!c->idx.empty()) {
        std::copy(c->idx.begin(), c->idx.end(), idx);
        std::copy(c->val.begin(), c->val.end(), val);
    }
    return mgbsetup_csr_sort_rows(c->rows, ptr, idx, val);
}

// Row sums of a CSR matrix exactly as `np.add.reduceat(data, indptr[nonempty])` (what scipy's `M.sum(axis=1)` runs) forms
// them: first entry + pairwise_sum(rest), numpy's pairwise_sum being a plain loop below 8 terms, eight running sums up to 128,
// and a split above (numpy/_core/src/umath/loops_utils.h.src).  The all-ones `uniform` subspace of a ladder is such a row sum.
static double np_pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

int mgbsetup_csr_row_sums(int64_t rows, const int32_t* ptr, const double* val, double* out) {
    if (rows < 0 || (rows > 0 && (!ptr || !out))) return 1;
    for (int64_t i = 0; i < rows; ++i) {
        const int64_t lo = ptr[i], n = ptr[i + 1] - lo;
        out[i] = n == 0 ? 0.0 : val[lo] + np_pairwise_sum(val + lo + 1, n - 1);
    }
    return 0;
}

struct MgbSetupCsr { int64_t rows, cols; const int32_t* ptr; const int32_t* idx; const double* val; };
struct MgbSetupOut { int64_t nnz; int32_t* ptr; int32_t* idx; double* val; };

// Block-diagonal concatenation of CSR matrices (AMG.R_fine = blockdiag of the state variables' prolongators,
// src/multigrid.jl:491) into caller-owned arrays: row pointers shifted by the entries before, columns by the columns before.
int mgbsetup_blockdiag(int32_t nblocks, const MgbSetupCsr* blk, int32_t* ptr, int32_t* idx, double* val) {
    if (nblocks < 0 || (nblocks > 0 && (!blk || !ptr))) return 1;
    int64_t r = 0, nz = 0, c0 = 0;
    ptr[0] = 0;
    for (int32_t b = 0; b < nblocks; ++b) {
        const MgbSetupCsr& B = blk[b];
        const int64_t bn = B.ptr[B.rows];
        if (nz + bn >= (int64_t)INT32_MAX || c0 + B.cols >= (int64_t)INT32_MAX) return 2;
        for (int64_t i = 0; i < B.rows; ++i) ptr[r + i + 1] = (int32_t)(nz + B.ptr[i + 1]);
        for (int64_t q = 0; q < bn; ++q) idx[nz + q] = (int32_t)(c0 + B.idx[q]);
        if (bn > 0) std::memcpy(val + nz, B.val, sizeof(double) * (size_t)bn);
        r += B.rows; nz += bn; c0 += B.cols;
    }
    return 0;
}

// The whole ladder in one call: product k+1 is formed while product k is copied out and sorted on a second thread (both only
// read product k).  outs[k] receives malloc'ed arrays (ptr: rows + 1, idx / val: nnz[k]) that the caller releases with
// mgbsetup_free.  Returns 0, or the negative code of mgbsetup_chain_multiply.

void mgbsetup_free(void* p) { std::free(p); }

int64_t mgbsetup_chain_run(void* h, int32_t nfac, const MgbSetupCsr* fac, MgbSetupOut* outs) {
    Chain* c = static_cast<Chain*>(h);
    if (!c || nfac < 0 || (nfac > 0 && (!fac || !outs))) return -1;
    std::vector<std::unique_ptr<Chain>> prods;           // product k stays alive while its emitter and product k + 1 read it
    std::vector<std::thread> emitters;
    int64_t rc = 0;
    const Chain* prev = c;
    for (int32_t k = 0; k < nfac; ++k) {
        if (fac[k].rows != prev->cols || fac[k].cols < 0 || !fac[k].ptr) { rc = -1; break; }
        prods.emplace_back(new Chain());
        Chain* cur = prods.back().get();
        if (!chain_product(*prev, fac[k].cols, fac[k].ptr, fac[k].idx, fac[k].val, *cur)) { rc = -2; break; }
        const int64_t nnz = full_nnz(*cur);
        if (nnz >= (int64_t)INT32_MAX) { rc = -2; break; }
        MgbSetupOut* o = &outs[k];
        o->nnz = nnz;
        o->ptr = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (size_t)(cur->full_rows + 1)));
        o->idx = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)));
        o->val = static_cast<double*>(std::malloc(sizeof(double) * (size_t)std::max<int64_t>(nnz, 1)));
        if (!o->ptr || !o->idx || !o->val) { rc = -3; break; }
        emitters.emplace_back([cur, o] { mgbsetup_chain_emit_sorted(cur, o->ptr, o->idx, o->val); });
        prev = cur;
    }
    for (auto& t : emitters) t.join();
    if (rc == 0 && !prods.empty()) *c = std::move(*prods.back());
    return rc;
}

}  // extern "C"
