// setup_host.cpp -- libmgbsetup.so: host-side helpers of the Python setup layer (row f4, time to first solution).
// Plain C++ (g++, no HIP): the two sequential loops of the AMG hierarchy construction that NumPy cannot vectorise.
// Each has a pure-Python twin in the package that tests/test_setup.py compares bit for bit; the package falls back to
// the twin when this library has not been built (the setup layer is host code either way -- the Newton path has no fallback).
#include <algorithm>
#include <cstdint>
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

}  // extern "C"
