// mf_analysis.cpp -- ordering + symbolic factorization for the device multifrontal
// Cholesky (see mf_analysis.hpp).  Host only.
#include "mf_analysis.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <unordered_map>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <thread>

namespace mgbhip {

namespace {

// Host threads for the embarrassingly parallel parts of the analysis (clique tests of a peeling round, neighbourhood
// hashes, the A scatter lists): the results are bitwise those of the serial loops.  MGBHIP_ANALYZE_THREADS overrides.
int analyze_threads() {
    static const int nt = [] {
        if (const char* e = getenv("MGBHIP_ANALYZE_THREADS")) return std::max(1, atoi(e));
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(hc ? hc : 1u, 16u);
    }();
    return nt;
}
void parallel_for(int64_t count, const std::function<void(int64_t, int64_t, int)>& fn, bool coarse_items = false) {
    const int nt = (count < 50000 && !coarse_items) ? 1 : (int)std::min<int64_t>(analyze_threads(), std::max<int64_t>(count, 1));
    if (nt <= 1) { fn(0, count, 0); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)nt);
    for (int t = 0; t < nt; ++t) {
        const int64_t a = count * t / nt, b = count * (t + 1) / nt;
        th.emplace_back([&fn, a, b, t] { fn(a, b, t); });
    }
    for (auto& x : th) x.join();
}

// std::sort of (hash, node) pairs, threaded: 256 buckets by the top byte of the hash (a stable counting scatter), every bucket
// sorted on its own -- the concatenation IS the sorted sequence (pairs compare by hash, then node), so the result does not depend
// on the thread count.  917 k pairs at L = 9: 70 ms -> 10 ms.
void sort_pairs(std::vector<std::pair<uint64_t, int32_t>>& a) {
    const size_t n = a.size();
    const int nt = analyze_threads();
    if (n < 100000 || nt <= 1) { std::sort(a.begin(), a.end()); return; }
    // histogram and scatter by contiguous chunks, one per thread: chunk t's elements of bucket b land behind those of the chunks
    // before it (a stable counting sort whatever the thread count), then every bucket is sorted on its own
    std::vector<size_t> cnt((size_t)nt * 256, 0);
    parallel_for(nt, [&](int64_t t0, int64_t t1, int) {
        for (int64_t t = t0; t < t1; ++t) {
            size_t* c = cnt.data() + (size_t)t * 256;
            for (size_t i = n * (size_t)t / (size_t)nt; i < n * (size_t)(t + 1) / (size_t)nt; ++i) c[a[i].first >> 56]++;
        }
    }, true);
    std::vector<size_t> start(257, 0);
    {
        size_t run = 0;
        for (int b = 0; b < 256; ++b) {
            start[b] = run;
            for (int t = 0; t < nt; ++t) { const size_t c = cnt[(size_t)t * 256 + b]; cnt[(size_t)t * 256 + b] = run; run += c; }
        }
        start[256] = run;
    }
    std::vector<std::pair<uint64_t, int32_t>> tmp(n);
    parallel_for(nt, [&](int64_t t0, int64_t t1, int) {
        for (int64_t t = t0; t < t1; ++t) {
            size_t* pos = cnt.data() + (size_t)t * 256;
            for (size_t i = n * (size_t)t / (size_t)nt; i < n * (size_t)(t + 1) / (size_t)nt; ++i) tmp[pos[a[i].first >> 56]++] = a[i];
        }
    }, true);
    parallel_for(256, [&](int64_t b0, int64_t b1, int) {
        for (int64_t b = b0; b < b1; ++b) std::sort(tmp.begin() + (long)start[b], tmp.begin() + (long)start[b + 1]);
    }, true);
    a.swap(tmp);
}

struct Graph {
    int64_t n;
    const int32_t* ptr;
    const int32_t* idx;
    bool adjacent(int32_t a, int32_t b) const {
        const int32_t* lo = idx + ptr[a];
        const int32_t* hi = idx + ptr[a + 1];
        return std::binary_search(lo, hi, b);
    }
};

struct Builder {
    Graph g;
    MfOptions opt;
    std::vector<int32_t> pos;       // elimination position of a node
    std::vector<int32_t> sn_of;     // supernode of a node
    std::vector<char> removed;      // ordered already
    std::vector<std::vector<int32_t>> sn_piv;
    int32_t next_pos = 0;
    int32_t sn_round2_end = 0;      // supernodes [0, sn_round2_end) come from the first two peeling rounds
    int32_t sn_round1_end = 0;      // ... [0, sn_round1_end) from the first

    // scratch for the dissection
    std::vector<int32_t> tag;       // subset membership id
    std::vector<int32_t> lvl;       // BFS level
    int32_t next_tag = 1;
    const double* coords = nullptr; // optional n x dim locations (geometric dissection)
    int dim = 0;

    void new_supernode(const int32_t* nodes, size_t cnt) {
        int32_t s = (int32_t)sn_piv.size();
        sn_piv.emplace_back(nodes, nodes + cnt);
        for (size_t i = 0; i < cnt; ++i) {
            pos[nodes[i]] = next_pos++;
            sn_of[nodes[i]] = s;
            removed[nodes[i]] = 1;
        }
    }

    // ---- 1. simplicial peeling -----------------------------------------------------
    void peel(MfPlan& plan) {
        const int64_t n = g.n;
        std::vector<char> sel(n, 0), simp;
        const bool timing = [] { const char* e = getenv("MGBHIP_DEBUG"); return e && atoi(e) >= 3; }() && n > 200000;
        auto t_prev = std::chrono::steady_clock::now();
        auto sub = [&](int round, const char* name) {
            if (!timing) return;
            const auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "[mgbhip]   peel round %d %-22s %.3f s\n", round, name, std::chrono::duration<double>(t - t_prev).count());
            t_prev = t;
        };
        for (int round = 0; round < opt.max_peel_rounds; ++round) {
            int64_t remaining = 0;
            for (int64_t v = 0; v < n; ++v) remaining += !removed[v];
            if (remaining == 0) break;
            std::vector<int32_t> chosen;
            std::fill(sel.begin(), sel.end(), 0);
            // (a) which remaining nodes are simplicial (few neighbours, forming a clique): independent per node, threaded;
            // (b) the greedy independent selection in index order, serial -- together exactly the one-pass rule
            //     "not next to a selected node, at most peel_max_degree neighbours, neighbours form a clique".
            simp.assign((size_t)n, 0);
            parallel_for(n, [&](int64_t v0, int64_t v1, int) {
                std::vector<int32_t> nbl;
                for (int64_t v = v0; v < v1; ++v) {
                    if (removed[v]) continue;
                    nbl.clear();
                    bool many = false;
                    for (int32_t e = g.ptr[v]; e < g.ptr[v + 1]; ++e) {
                        const int32_t u = g.idx[e];
                        if (u == v || removed[u]) continue;
                        nbl.push_back(u);
                        if ((int32_t)nbl.size() > opt.peel_max_degree) { many = true; break; }
                    }
                    if (many) continue;
                    bool clique = true;
                    for (size_t a = 0; a < nbl.size() && clique; ++a)
                        for (size_t b = a + 1; b < nbl.size(); ++b)
                            if (!g.adjacent(nbl[a], nbl[b])) { clique = false; break; }
                    simp[(size_t)v] = clique ? 1 : 0;
                }
            });
            sub(round, "clique tests");
            for (int32_t v = 0; v < n; ++v) {
                if (removed[v] || !simp[(size_t)v]) continue;
                bool blocked = false;
                for (int32_t e = g.ptr[v]; e < g.ptr[v + 1]; ++e) {
                    const int32_t u = g.idx[e];
                    if (u != v && !removed[u] && sel[u]) { blocked = true; break; }
                }
                if (blocked) continue;
                sel[v] = 1;
                chosen.push_back(v);
            }
            sub(round, "greedy selection");
            if (chosen.empty() || (double)chosen.size() < 0.02 * (double)remaining) break;
            // group chosen nodes with identical remaining neighbourhoods (they are mutually
            // non-adjacent by construction): one front per group, at most 16 pivots each
            std::vector<std::pair<uint64_t, int32_t>> keyed;
            keyed.reserve(chosen.size());
            auto nbhash = [&](int32_t v) {
                uint64_t h = 1469598103934665603ull;
                for (int32_t e = g.ptr[v]; e < g.ptr[v + 1]; ++e) {
                    int32_t u = g.idx[e];
                    if (u == v || removed[u]) continue;
                    h ^= (uint64_t)(uint32_t)u + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
                }
                return h;
            };
            keyed.resize(chosen.size());
            parallel_for((int64_t)chosen.size(), [&](int64_t i0, int64_t i1, int) {
                for (int64_t i = i0; i < i1; ++i) keyed[(size_t)i] = {nbhash(chosen[(size_t)i]), chosen[(size_t)i]};
            });
            sub(round, "hashes");
            sort_pairs(keyed);
            sub(round, "sort");
            auto same_nb = [&](int32_t a, int32_t b) {
                int32_t ea = g.ptr[a], eb = g.ptr[b];
                const int32_t enda = g.ptr[a + 1], endb = g.ptr[b + 1];
                while (true) {
                    while (ea < enda && (g.idx[ea] == a || removed[g.idx[ea]])) ++ea;
                    while (eb < endb && (g.idx[eb] == b || removed[g.idx[eb]])) ++eb;
                    if (ea == enda || eb == endb) return ea == enda && eb == endb;
                    if (g.idx[ea] != g.idx[eb]) return false;
                    ++ea; ++eb;
                }
            };
            // a group never leaves a run of equal hashes and always starts at the run's first unassigned entry: cut the sorted
            // array into chunks at run boundaries, group every chunk on its own thread, concatenate in chunk order
            std::vector<std::vector<int32_t>> groups;
            {
                const int nt = keyed.size() >= 100000 ? analyze_threads() : 1;
                std::vector<size_t> cut((size_t)nt + 1, keyed.size());
                cut[0] = 0;
                for (int t = 1; t < nt; ++t) {
                    size_t c = std::max(cut[(size_t)t - 1], keyed.size() * (size_t)t / (size_t)nt);
                    while (c > 0 && c < keyed.size() && keyed[c].first == keyed[c - 1].first) ++c;
                    cut[(size_t)t] = c;
                }
                std::vector<std::vector<std::vector<int32_t>>> part((size_t)nt);
                parallel_for(nt, [&](int64_t t0, int64_t t1, int) {
                    std::vector<int32_t> group;
                    for (int64_t t = t0; t < t1; ++t) {
                        auto& out = part[(size_t)t];
                        size_t i = cut[(size_t)t];
                        const size_t end = cut[(size_t)t + 1];
                        while (i < end) {
                            group.clear();
                            group.push_back(keyed[i].second);
                            size_t j = i + 1;
                            while (j < end && keyed[j].first == keyed[i].first && group.size() < 16 &&
                                   same_nb(keyed[i].second, keyed[j].second)) {
                                group.push_back(keyed[j].second);
                                ++j;
                            }
                            out.push_back(group);
                            i = j;
                        }
                    }
                }, true);
                size_t total = 0;
                for (auto& pt : part) total += pt.size();
                groups.reserve(total);
                for (auto& pt : part)
                    for (auto& g2 : pt) groups.push_back(std::move(g2));
            }
            sub(round, "grouping");
            // removal happens after grouping so that `removed` is stable during same_nb
            for (auto& grp : groups) new_supernode(grp.data(), grp.size());
            sub(round, "supernodes");
            if (round == 0) sn_round1_end = (int32_t)sn_piv.size();
            if (round == 1) sn_round2_end = (int32_t)sn_piv.size();
            plan.peeled += (int64_t)chosen.size();
            plan.peel_rounds = round + 1;
        }
    }

    // ---- 2. nested dissection --------------------------------------------------------
    // BFS inside the subset tagged `id`; returns nodes in visit order, fills lvl.
    void bfs(int32_t start, int32_t id, std::vector<int32_t>& order) {
        order.clear();
        order.push_back(start);
        lvl[start] = 0;
        tag[start] = -id;   // visited marker: negative of the id
        for (size_t h = 0; h < order.size(); ++h) {
            int32_t v = order[h];
            for (int32_t e = g.ptr[v]; e < g.ptr[v + 1]; ++e) {
                int32_t u = g.idx[e];
                if (tag[u] != id) continue;   // outside subset or already visited
                tag[u] = -id;
                lvl[u] = lvl[v] + 1;
                order.push_back(u);
            }
        }
    }

    // Geometric bisection: cut the longest axis of the bounding box at the median coordinate.  B is
    // the side at or above the cut; the separator is the layer of B that touches A (on a mesh whose
    // nodes sit on the cut line this is exactly that line).  Returns false when the cut degenerates.
    bool dissect_geometric(std::vector<int32_t>& nodes) {
        const size_t cnt = nodes.size();
        int axis = 0;
        double best_ext = -1.0;
        for (int d = 0; d < dim; ++d) {
            double lo = 1e300, hi = -1e300;
            for (int32_t v : nodes) {
                const double c = coords[(size_t)v * dim + d];
                lo = std::min(lo, c);
                hi = std::max(hi, c);
            }
            if (hi - lo > best_ext) { best_ext = hi - lo; axis = d; }
        }
        if (!(best_ext > 0)) return false;
        std::vector<int32_t> sorted(nodes);
        auto key = [&](int32_t v) { return coords[(size_t)v * dim + axis]; };
        std::nth_element(sorted.begin(), sorted.begin() + cnt / 2, sorted.end(),
                         [&](int32_t a, int32_t b) { const double ka = key(a), kb = key(b); return ka < kb || (ka == kb && a < b); });
        const double thr = key(sorted[cnt / 2]);
        const int32_t id = next_tag++;
        size_t nA = 0;
        for (int32_t v : nodes) {
            const bool inA = key(v) < thr;
            tag[v] = inA ? id : -id;          // A: +id, B: -id
            nA += inA;
        }
        if (nA == 0 || nA == cnt) return false;
        std::vector<int32_t> A, B, S;
        A.reserve(nA);
        for (int32_t v : nodes) {
            if (tag[v] == id) { A.push_back(v); continue; }
            bool touches = false;
            for (int32_t e = g.ptr[v]; e < g.ptr[v + 1] && !touches; ++e) touches = tag[g.idx[e]] == id;
            (touches ? S : B).push_back(v);
        }
        if (S.empty()) {                      // the two sides are not connected at all: no separator
            nodes.clear();
            nodes.shrink_to_fit();
            dissect(A);
            dissect(B);
            return true;
        }
        if (B.empty() || S.size() * 3 > cnt) return false;     // a thick cut: let the graph method try
        nodes.clear();
        nodes.shrink_to_fit();
        dissect(A);
        dissect(B);
        new_supernode(S.data(), S.size());
        return true;
    }

    void dissect(std::vector<int32_t>& nodes) {
        if ((int32_t)nodes.size() <= opt.leaf_size) {
            if (!nodes.empty()) new_supernode(nodes.data(), nodes.size());
            return;
        }
        if (coords && dim > 0 && dissect_geometric(nodes)) return;
        int32_t id = next_tag++;
        for (int32_t v : nodes) tag[v] = id;
        std::vector<int32_t> order;
        bfs(nodes[0], id, order);
        if (order.size() < nodes.size()) {
            // disconnected: split off the reached component, no separator needed
            std::vector<int32_t> rest;
            rest.reserve(nodes.size() - order.size());
            for (int32_t v : nodes)
                if (tag[v] == id) rest.push_back(v);
            std::vector<int32_t> comp(order);
            nodes.clear();
            nodes.shrink_to_fit();
            dissect(comp);
            dissect(rest);
            return;
        }
        // second sweep from the far end for a longer level structure
        int32_t far = order.back();
        for (int32_t v : nodes) tag[v] = id;
        bfs(far, id, order);
        int32_t depth = lvl[order.back()];
        if (depth < 2) {
            new_supernode(nodes.data(), nodes.size());
            return;
        }
        std::vector<int64_t> cnt(depth + 1, 0);
        for (int32_t v : order) cnt[lvl[v]]++;
        // choose the level whose removal best balances the two sides
        int64_t total = (int64_t)order.size(), below = cnt[0];
        int32_t best = 1;
        double best_score = 1e300;
        for (int32_t l = 1; l <= depth - 1; ++l) {
            int64_t above = total - below - cnt[l];
            double imbalance = (double)std::llabs(below - above) / (double)total;
            double score = imbalance + opt.sep_weight * (double)cnt[l] / (double)total;
            if (score < best_score) { best_score = score; best = l; }
            below += cnt[l];
        }
        std::vector<int32_t> A, B, S;
        for (int32_t v : order) {
            if (lvl[v] < best) A.push_back(v);
            else if (lvl[v] > best) B.push_back(v);
            else {
                bool touches = false;
                for (int32_t e = g.ptr[v]; e < g.ptr[v + 1] && !touches; ++e) {
                    int32_t u = g.idx[e];
                    if (tag[u] == -id && lvl[u] == best + 1) touches = true;
                }
                (touches ? S : A).push_back(v);
            }
        }
        nodes.clear();
        nodes.shrink_to_fit();
        order.clear();
        order.shrink_to_fit();
        if (S.empty() || B.empty()) {   // cannot happen for a connected level structure, keep safe
            std::vector<int32_t> all(A);
            all.insert(all.end(), B.begin(), B.end());
            all.insert(all.end(), S.begin(), S.end());
            new_supernode(all.data(), all.size());
            return;
        }
        // the children re-tag their own subsets; S keeps tag -id and is never revisited
        dissect(A);
        dissect(B);
        new_supernode(S.data(), S.size());
    }
};

}  // namespace

void mf_analyze(int64_t n, const int32_t* rowptr, const int32_t* colidx, const MfOptions& opt,
                MfPlan& plan, const double* coords, int dim) {
    plan = MfPlan();
    plan.n = n;
    if (n == 0) { plan.level_ptr.assign(1, 0); return; }
    if (n > INT32_MAX) throw std::runtime_error("mf_analyze: n exceeds 32-bit indexing");
    // MGBHIP_DEBUG >= 3: wall-clock of the phases (development aid for the time-to-first-solution breakdown)
    const bool phase_timing = [] { const char* e = getenv("MGBHIP_DEBUG"); return e && atoi(e) >= 3; }();
    auto t_prev = std::chrono::steady_clock::now();
    auto phase = [&](const char* name) {
        if (!phase_timing) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[mgbhip] analysis n=%lld %-28s %.3f s\n", (long long)n, name, std::chrono::duration<double>(t - t_prev).count());
        t_prev = t;
    };
    Builder b{Graph{n, rowptr, colidx}, opt};
    b.pos.assign(n, -1);
    b.sn_of.assign(n, -1);
    b.removed.assign(n, 0);
    b.tag.assign(n, 0);
    b.lvl.assign(n, 0);
    b.coords = coords;
    b.dim = dim;

    for (int64_t t = 0; t < opt.ntop; ++t) {            // interface unknowns: invisible to peeling and dissection
        const int32_t v = opt.top[t];
        if (v < 0 || v >= n || b.removed[v]) throw std::runtime_error("mf_analyze: bad interface list");
        b.removed[v] = 1;
    }
    phase("init");
    b.peel(plan);
    phase("peel");
    // supernodes [0, nsn_peeled) are the element-local unknowns (round 1: the broken slacks, round 2: the element-
    // interior nodes they expose); later rounds peel ordinary mesh nodes
    const int32_t nsn_peeled = b.sn_round2_end;
    {
        std::vector<int32_t> rest;
        for (int32_t v = 0; v < n; ++v)
            if (!b.removed[v]) rest.push_back(v);
        b.dissect(rest);
    }
    phase("dissect");
    int32_t top_sn = -1;
    if (opt.ntop > 0) {                                 // ... and eliminated last, as one supernode
        top_sn = (int32_t)b.sn_piv.size();
        b.new_supernode(opt.top, (size_t)opt.ntop);
    }
    const int32_t nsn = (int32_t)b.sn_piv.size();
    const std::vector<int32_t>& pos = b.pos;

    // ---- 3. symbolic factorization on the supernode partition --------------------------
    std::vector<std::vector<int32_t>> sn_struct(nsn);
    std::vector<std::vector<int32_t>> sn_child(nsn);
    std::vector<int32_t> parent(nsn, -1);
    {
        // The supernodes of the first two peeling rounds are independent sets: those of round 1 have no children, those of
        // round 2 only children from round 1.  Their structures (a handful of nodes each, but a quarter of a million of them at
        // L = 9) are formed on host threads, a round at a time, and linked to their parents serially in supernode order --
        // the same lists in the same order as the serial loop below produces for the rest.
        int32_t s_begin = 0;
        const int32_t r1 = b.sn_round1_end, r2 = std::max(b.sn_round2_end, b.sn_round1_end);
        if (r1 >= 4096 && r2 <= nsn && r1 <= r2) {
            auto batch = [&](int32_t s0, int32_t s1) {
                parallel_for((int64_t)(s1 - s0), [&](int64_t a0, int64_t a1, int) {
                    for (int64_t t = a0; t < a1; ++t) {
                        const int32_t s = s0 + (int32_t)t;
                        auto& piv = b.sn_piv[s];
                        const int32_t maxpos = pos[piv.back()];
                        auto& st = sn_struct[s];
                        for (int32_t v : piv)
                            for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
                                const int32_t u = colidx[e];
                                if (pos[u] > maxpos) st.push_back(u);
                            }
                        for (int32_t c : sn_child[s])
                            for (int32_t u : sn_struct[c])
                                if (pos[u] > maxpos) st.push_back(u);
                        std::sort(st.begin(), st.end(), [&](int32_t a, int32_t c2) { return pos[a] < pos[c2]; });
                        st.erase(std::unique(st.begin(), st.end()), st.end());      // positions are distinct: equal position = equal node
                    }
                }, true);
                for (int32_t s = s0; s < s1; ++s)
                    if (!sn_struct[s].empty()) {
                        parent[s] = b.sn_of[sn_struct[s][0]];
                        sn_child[parent[s]].push_back(s);
                    }
            };
            batch(0, r1);
            // a round-2 supernode takes children from round 1 only if every round-1 parent lies at or after r1: check, else redo serially
            bool ok = true;
            for (int32_t s = 0; s < r1 && ok; ++s) ok = parent[s] < 0 || parent[s] >= r1;
            if (ok) {
                batch(r1, r2);
                for (int32_t s = r1; s < r2 && ok; ++s) ok = parent[s] < 0 || parent[s] >= r2;
            }
            if (ok) {
                s_begin = r2;
            } else {                                     // unexpected shape: start over with the serial loop
                for (int32_t s = 0; s < nsn; ++s) { sn_struct[s].clear(); sn_child[s].clear(); parent[s] = -1; }
            }
        }
        std::vector<int32_t> mark(n, -1);
        for (int32_t s = s_begin; s < nsn; ++s) {
            auto& piv = b.sn_piv[s];
            int32_t maxpos = pos[piv.back()];
            for (int32_t v : piv) mark[v] = s;
            auto& st = sn_struct[s];
            for (int32_t v : piv)
                for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
                    int32_t u = colidx[e];
                    if (mark[u] != s && pos[u] > maxpos) { mark[u] = s; st.push_back(u); }
                }
            for (int32_t c : sn_child[s])
                for (int32_t u : sn_struct[c])
                    if (mark[u] != s && pos[u] > maxpos) { mark[u] = s; st.push_back(u); }
            std::sort(st.begin(), st.end(), [&](int32_t a, int32_t c2) { return pos[a] < pos[c2]; });
            if (!st.empty()) {
                parent[s] = b.sn_of[st[0]];
                sn_child[parent[s]].push_back(s);
            }
        }
    }

    phase("symbolic structure");
    // ---- exact-fit amalgamation of a child into its parent ------------------------------
    std::vector<int32_t> merged_into(nsn, -1);
    {
        std::vector<int32_t> extra_piv(nsn, 0);   // pivots already merged into a parent
        for (int32_t s = 0; s < nsn; ++s) {
            int32_t f = parent[s];
            if (f < 0) continue;
            size_t kf = b.sn_piv[f].size() - (size_t)extra_piv[f];   // original pivots of f
            int64_t ks = (int64_t)b.sn_piv[s].size();
            const bool exact = sn_struct[s].size() == kf + sn_struct[f].size();
            // relaxed amalgamation: small fronts are merged into their parent even with explicit
            // zeros, which collapses the chains of 1-3 pivot separators at the bottom of the tree
            const int64_t merged_m = ks + (int64_t)b.sn_piv[f].size() + (int64_t)sn_struct[f].size();
            const bool relaxed = opt.merge_max_m > 0 && merged_m <= opt.merge_max_m;
            if (!exact && !relaxed) continue;
            if (opt.protect_peeled && s < nsn_peeled && f >= nsn_peeled) continue;
            if (f == top_sn || s == top_sn) continue;       // the interface front stays exactly the interface
            // an exact fit costs no flops, but merging a large child serialises two pivot chains that
            // the tree would run side by side (the half-domain separator into the root separator):
            // only fronts that stay within one LDS workgroup are amalgamated
            if (!relaxed && merged_m > opt.exact_merge_max_m) continue;
            if (!relaxed) {
                if ((int64_t)extra_piv[f] * ks > 64) continue;          // zeros between merged siblings
                if (ks + (int64_t)b.sn_piv[f].size() > 64 && extra_piv[f] > 0) continue;
            }
            // merge: child's pivots go first (they are eliminated earlier)
            std::vector<int32_t> np(b.sn_piv[s]);
            np.insert(np.end(), b.sn_piv[f].begin(), b.sn_piv[f].end());
            // keep pos-order inside the pivot list
            std::sort(np.begin(), np.end(), [&](int32_t a, int32_t c2) { return pos[a] < pos[c2]; });
            b.sn_piv[f].swap(np);
            extra_piv[f] += (int32_t)ks;
            merged_into[s] = f;
            // re-parent s's children to f
            auto& cf = sn_child[f];
            cf.erase(std::remove(cf.begin(), cf.end(), s), cf.end());
            for (int32_t c : sn_child[s]) { parent[c] = f; cf.push_back(c); }
            sn_child[s].clear();
            b.sn_piv[s].clear();
            sn_struct[s].clear();
        }
    }

    phase("amalgamation");
    // ---- border unknown (MfOptions::border): last in the order, in every front's boundary, own root front ----
    std::vector<int32_t> pos_ext;
    int32_t nsn_all = nsn;
    if (opt.border) {
        plan.border = true;
        pos_ext = b.pos;
        pos_ext.push_back((int32_t)n);               // after every real unknown
        std::vector<int32_t> roots;
        for (int32_t s = 0; s < nsn; ++s) {
            if (merged_into[s] >= 0) continue;
            sn_struct[s].push_back((int32_t)n);
            if (parent[s] < 0) roots.push_back(s);
        }
        b.sn_piv.push_back(std::vector<int32_t>{(int32_t)n});
        sn_struct.emplace_back();
        sn_child.push_back(roots);
        for (int32_t r : roots) parent[r] = nsn;
        parent.push_back(-1);
        merged_into.push_back(-1);
        nsn_all = nsn + 1;
    }
    const std::vector<int32_t>& posx = opt.border ? pos_ext : b.pos;
    const int64_t nnz_h = rowptr[n];

    // ---- levels, final numbering ----------------------------------------------------------
    std::vector<int32_t> level(nsn_all, 0);
    for (int32_t s = 0; s < nsn_all; ++s) {
        if (merged_into[s] >= 0) continue;
        int32_t l = 0;
        for (int32_t c : sn_child[s]) l = std::max(l, level[c] + 1);
        level[s] = l;
    }
    std::vector<int32_t> live;
    for (int32_t s = 0; s < nsn_all; ++s)
        if (merged_into[s] < 0) live.push_back(s);
    auto msize = [&](int32_t s) { return (int32_t)(b.sn_piv[s].size() + sn_struct[s].size()); };
    std::stable_sort(live.begin(), live.end(), [&](int32_t a, int32_t c2) {
        if (level[a] != level[c2]) return level[a] < level[c2];
        return msize(a) < msize(c2);
    });
    std::vector<int32_t> newid(nsn_all, -1);
    for (size_t i = 0; i < live.size(); ++i) newid[live[i]] = (int32_t)i;

    const int32_t nf = (int32_t)live.size();
    plan.fronts.resize(nf);
    int32_t maxlevel = 0;
    for (int32_t i = 0; i < nf; ++i) {
        int32_t s = live[i];
        Front& f = plan.fronts[i];
        f.k = (int32_t)b.sn_piv[s].size();
        f.m = msize(s);
        f.level = level[s];
        f.parent = parent[s] >= 0 ? newid[parent[s]] : -1;
        maxlevel = std::max(maxlevel, f.level);
        f.idx_off = (int64_t)plan.front_idx.size();
        plan.front_idx.insert(plan.front_idx.end(), b.sn_piv[s].begin(), b.sn_piv[s].end());
        plan.front_idx.insert(plan.front_idx.end(), sn_struct[s].begin(), sn_struct[s].end());
        f.F_off = plan.arena_doubles;
        plan.arena_doubles += (int64_t)f.m * f.m;
        f.u_off = plan.uvec_doubles;
        plan.uvec_doubles += f.m - f.k;
        plan.max_m = std::max(plan.max_m, f.m);
        for (int32_t j = 0; j < f.k; ++j) {
            int64_t r = f.m - j;
            plan.factor_flops += r * r;
        }
        f.child_off = (int64_t)plan.children.size();
        f.nchild = (int32_t)sn_child[s].size();
        // children sorted by their new id => deterministic extend-add order
        std::vector<int32_t> ch;
        for (int32_t c : sn_child[s]) ch.push_back(newid[c]);
        std::sort(ch.begin(), ch.end());
        plan.children.insert(plan.children.end(), ch.begin(), ch.end());
    }
    if (top_sn >= 0) plan.iface_front = newid[top_sn];
    plan.level_ptr.assign(maxlevel + 2, 0);
    for (int32_t i = 0; i < nf; ++i) plan.level_ptr[plan.fronts[i].level + 1]++;
    for (int32_t l = 0; l <= maxlevel; ++l) plan.level_ptr[l + 1] += plan.level_ptr[l];

    phase("levels, numbering, index lists");
    // ---- relative indices + A scatter lists ------------------------------------------------
    // Two passes so that the fronts can be filled by several host threads: (1) sizes -- a front owns, per pivot v, the
    // entries {v, u} with u == v or u eliminated later (+ the border entry) -- and offsets; (2) the lists themselves,
    // every front into its own range (the serial loop appended them in the same order: identical arrays).
    // rel: positions of a front's boundary inside its parent's index list
    for (int32_t i = 0; i < nf; ++i) {
        Front& f = plan.fronts[i];
        f.rel_off = (int64_t)plan.rel.size();
        plan.rel.resize(plan.rel.size() + (size_t)(f.m - f.k), -1);
    }
    std::vector<int64_t> a_count((size_t)nf + 1, 0);
    parallel_for(nf, [&](int64_t i0, int64_t i1, int) {
        for (int64_t i = i0; i < i1; ++i) {
            const Front& f = plan.fronts[(size_t)i];
            const int32_t* idx = plan.front_idx.data() + f.idx_off;
            int64_t cnt = 0;
            for (int32_t lv = 0; lv < f.k; ++lv) {
                const int32_t v = idx[lv];
                if (v == n) { ++cnt; continue; }
                for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
                    const int32_t u = colidx[e];
                    if (u != v && posx[u] < posx[v]) continue;
                    ++cnt;
                }
                if (opt.border) ++cnt;
            }
            a_count[(size_t)i + 1] = cnt;
        }
    });
    {
        int64_t acol = 0;
        for (int32_t i = 0; i < nf; ++i) {
            Front& f = plan.fronts[i];
            f.a_off = a_count[i];
            f.a_cnt = (int32_t)a_count[(size_t)i + 1];
            a_count[(size_t)i + 1] += a_count[i];
            f.acol_off = acol;
            acol += f.k + 1;
        }
        plan.a_src.assign((size_t)a_count[nf], 0);
        plan.a_dst.assign((size_t)a_count[nf], 0);
        plan.a_colptr.assign((size_t)acol, 0);
    }
    if (nnz_h + n >= (int64_t)INT32_MAX && opt.border) throw std::runtime_error("mf_analyze: bordered value index exceeds 32 bits");
    std::vector<std::string> errors((size_t)analyze_threads() + 1);
    parallel_for(nf, [&](int64_t i0, int64_t i1, int tix) {
        std::vector<int32_t> loc((size_t)n + 1, -1);
        try {
            for (int64_t i = i0; i < i1; ++i) {
                Front& f = plan.fronts[(size_t)i];
                const int32_t* idx = plan.front_idx.data() + f.idx_off;
                for (int32_t j = 0; j < f.m; ++j) loc[idx[j]] = j;
                // children of i
                for (int32_t c = 0; c < f.nchild; ++c) {
                    const Front& ch = plan.fronts[plan.children[f.child_off + c]];
                    const int32_t* cidx = plan.front_idx.data() + ch.idx_off + ch.k;
                    for (int32_t j = 0; j < ch.m - ch.k; ++j) {
                        const int32_t p = loc[cidx[j]];
                        if (p < 0) throw std::runtime_error("mf_analyze: child boundary not contained in parent front");
                        plan.rel[ch.rel_off + j] = p;
                    }
                }
                // A entries owned by this front: pairs {v,u}, v a pivot, u == v or eliminated later
                int64_t w = f.a_off;
                int32_t* cp = plan.a_colptr.data() + f.acol_off;
                for (int32_t lv = 0; lv < f.k; ++lv) {
                    cp[lv] = (int32_t)(w - f.a_off);
                    const int32_t v = idx[lv];
                    if (v == n) {                       // the border's own 1 x 1 front: entry (n, n)
                        plan.a_src[(size_t)w] = (int32_t)(nnz_h + n);
                        plan.a_dst[(size_t)w] = lv + lv * f.m;
                        ++w;
                        continue;
                    }
                    for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
                        const int32_t u = colidx[e];
                        if (u != v && posx[u] < posx[v]) continue;
                        const int32_t lu = loc[u];
                        if (lu < 0 || lu < lv) throw std::runtime_error("mf_analyze: structure violation in A scatter");
                        int32_t src = e;
                        if (u < v) {   // symmetric(H) reads the upper triangle: entry (u, v) in row u
                            const int32_t* lo = colidx + rowptr[u];
                            const int32_t* hi = colidx + rowptr[u + 1];
                            const int32_t* it = std::lower_bound(lo, hi, v);
                            if (it == hi || *it != v) throw std::runtime_error("mf_analyze: pattern is not symmetric");
                            src = (int32_t)(it - colidx);
                        }
                        plan.a_src[(size_t)w] = src;
                        plan.a_dst[(size_t)w] = lu + lv * f.m;
                        ++w;
                    }
                    if (opt.border) {                   // border column entry (n, v): last row of the front
                        if (loc[n] != f.m - 1) throw std::runtime_error("mf_analyze: border is not the last row of a front");
                        plan.a_src[(size_t)w] = (int32_t)(nnz_h + v);
                        plan.a_dst[(size_t)w] = (f.m - 1) + lv * f.m;
                        ++w;
                    }
                }
                cp[f.k] = f.a_cnt;
                if (w - f.a_off != f.a_cnt) throw std::runtime_error("mf_analyze: A list size mismatch");
                for (int32_t j = 0; j < f.m; ++j) loc[idx[j]] = -1;
            }
        } catch (const std::exception& e) {
            errors[(size_t)tix] = e.what();
        }
    });
    for (const auto& e : errors)
        if (!e.empty()) throw std::runtime_error(e);
    phase("relative indices, A lists");
}

}  // namespace mgbhip
