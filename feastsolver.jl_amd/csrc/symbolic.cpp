// Ordering (nested dissection by BFS level structures), elimination tree,
// supernode detection, relaxed amalgamation and front/arena layout.
// See symbolic.hpp for the role of this phase on the hot path.
#include "symbolic.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "ordering.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <numeric>
#include <thread>

namespace feasthip {

namespace {

struct Graph {
    int n = 0;
    std::vector<int64_t> xadj;
    std::vector<int> adj;
};

// adjacency of A + A^T without the diagonal, neighbours ascending.  The transpose comes out of a counting pass with its
// lists already ascending; a column of A whose row indices ascend (every CSC matrix Julia or scipy builds) is then merged with
// its row in one linear pass, any other column is sorted first.
Graph symmetrize(int n, const int64_t* colptr, const int64_t* rowidx) {
    Graph g;
    g.n = n;
    // Fast path: a structurally symmetric pattern with strictly ascending row indices (what every symmetric or Hermitian pencil handed
    // over by Julia or scipy is) IS its own transpose: the adjacency is the pattern without the diagonal.  Checked column range by
    // column range on host threads (entry (i, j) must find j in column i by binary search); the general path below takes 10 ms of
    // serial transposition on C3, this one 2.
    {
        const int nth = n >= 20000 ? std::max(1, std::min(8, host_cpu_budget() / 2)) : 1;
        std::vector<char> bad(nth, 0);
        std::vector<int64_t> cnt(n + 1, 0);
        auto check = [&](int ti) {
            const int j0 = (int)((int64_t)n * ti / nth), j1 = (int)((int64_t)n * (ti + 1) / nth);
            for (int j = j0; j < j1 && !bad[ti]; ++j) {
                int64_t c = 0;
                for (int64_t t = colptr[j]; t < colptr[j + 1]; ++t) {
                    const int64_t i = rowidx[t];
                    if (t > colptr[j] && rowidx[t - 1] >= i) { bad[ti] = 1; break; }
                    if (i == j) continue;
                    if (!std::binary_search(rowidx + colptr[i], rowidx + colptr[i + 1], (int64_t)j)) { bad[ti] = 1; break; }
                    ++c;
                }
                cnt[j + 1] = c;
            }
        };
        run_parts(nth, check);
        bool sym = true;
        for (char b : bad) if (b) sym = false;
        if (sym) {
            g.xadj.assign(n + 1, 0);
            for (int j = 0; j < n; ++j) g.xadj[j + 1] = g.xadj[j] + cnt[j + 1];
            g.adj.resize(g.xadj[n]);
            auto fill = [&](int ti) {
                const int j0 = (int)((int64_t)n * ti / nth), j1 = (int)((int64_t)n * (ti + 1) / nth);
                for (int j = j0; j < j1; ++j) {
                    int64_t o = g.xadj[j];
                    for (int64_t t = colptr[j]; t < colptr[j + 1]; ++t) if (rowidx[t] != j) g.adj[o++] = (int)rowidx[t];
                }
            };
            run_parts(nth, fill);
            return g;
        }
    }
    std::vector<int64_t> tp(n + 1, 0);
    for (int j = 0; j < n; ++j)
        for (int64_t t = colptr[j]; t < colptr[j + 1]; ++t) tp[rowidx[t] + 1]++;
    for (int j = 0; j < n; ++j) tp[j + 1] += tp[j];
    std::vector<int> tr(tp[n]);                       // row i of A: the columns j with a_ij != 0, ascending
    {
        std::vector<int64_t> pos(tp.begin(), tp.end() - 1);
        for (int j = 0; j < n; ++j)
            for (int64_t t = colptr[j]; t < colptr[j + 1]; ++t) tr[pos[rowidx[t]]++] = j;
    }
    g.xadj.assign(n + 1, 0);
    // column ranges on host threads, each into its own list; concatenated in order (the result does not depend on the thread count)
    const int nth = n >= 20000 ? std::max(1, std::min(8, host_cpu_budget() / 2)) : 1;
    std::vector<std::vector<int>> part(nth);
    auto merge_range = [&](int ti) {
        const int j0 = (int)((int64_t)n * ti / nth), j1 = (int)((int64_t)n * (ti + 1) / nth);
        std::vector<int>& out = part[ti];
        out.reserve((size_t)(colptr[j1] - colptr[j0]) + (size_t)(tp[j1] - tp[j0]));
        std::vector<int> own;
        for (int j = j0; j < j1; ++j) {
            const int64_t c0 = colptr[j], c1 = colptr[j + 1];
            bool sorted = true;
            for (int64_t t = c0 + 1; t < c1 && sorted; ++t) sorted = rowidx[t - 1] < rowidx[t];
            own.resize(c1 - c0);
            for (int64_t t = c0; t < c1; ++t) own[t - c0] = (int)rowidx[t];
            if (!sorted) { std::sort(own.begin(), own.end()); own.erase(std::unique(own.begin(), own.end()), own.end()); }
            const int* a = own.data(); const int* ae = a + own.size();
            const int* r = tr.data() + tp[j]; const int* re = tr.data() + tp[j + 1];
            const size_t start = out.size();
            while (a < ae || r < re) {
                int v;
                if (r >= re || (a < ae && *a < *r)) v = *a++;
                else if (a >= ae || *r < *a) v = *r++;
                else { v = *a; ++a; ++r; }
                if (v != j && (out.size() == start || out.back() != v)) out.push_back(v);     // (duplicate entries of a column show up twice in its row)
            }
            g.xadj[j + 1] = (int64_t)(out.size() - start);       // length for now, offsets below
        }
    };
    run_parts(nth, merge_range);
    for (int j = 0; j < n; ++j) g.xadj[j + 1] += g.xadj[j];
    g.adj.resize(g.xadj[n]);
    {
        int64_t off = 0;
        for (int ti = 0; ti < nth; ++ti) { std::copy(part[ti].begin(), part[ti].end(), g.adj.begin() + off); off += (int64_t)part[ti].size(); }
    }
    return g;
}

void permute_graph(const Graph& g, const std::vector<int>& perm, const std::vector<int>& iperm, Graph& h) {      // (h's arrays are reused: the analysis permutes three times)
    h.n = g.n;
    h.xadj.assign(g.n + 1, 0);
    h.adj.resize(g.adj.size());
    for (int j = 0; j < g.n; ++j) h.xadj[j + 1] = h.xadj[j] + (g.xadj[perm[j] + 1] - g.xadj[perm[j]]);
    // columns are independent: ranges of them on host threads (the analysis calls this three times; 4-5 ms each on one core for C3)
    auto fill = [&](int j0, int j1) {
        for (int j = j0; j < j1; ++j) {
            int64_t o = h.xadj[j];
            for (int64_t t = g.xadj[perm[j]]; t < g.xadj[perm[j] + 1]; ++t) h.adj[o++] = iperm[g.adj[t]];      // (unsorted: no reader needs the order, and sorting 97 k short lists was most of the 2.5 ms a call took)
        }
    };
    const int nth = g.n >= 20000 ? 8 : 1;
    run_parts(nth, [&](int t) { fill((int)((int64_t)g.n * t / nth), (int)((int64_t)g.n * (t + 1) / nth)); });
}

// ---------------------------------------------------------------------------
// Nested dissection.  Separators are BFS level sets from pseudo-peripheral
// vertices, trimmed to the vertices that actually touch both sides.
// ---------------------------------------------------------------------------
class NestedDissection {
public:
    NestedDissection(const Graph& g, int leaf) : g_(g), leaf_(leaf), part_(g.n, -1), vis_(g.n, -1), lvl_(g.n, 0), order_(g.n, -1), loc_(g.n, -1) {}

    std::vector<int> run() {
        std::vector<int> all(g_.n);
        std::iota(all.begin(), all.end(), 0);
        t_start_ = std::chrono::steady_clock::now();
        dissect(all, 0);
        if (getenv("FEASTHIP_ANALYZE_TRACE")) {
            fprintf(stderr, "[feasthip nd] last separator of depth d found at (ms):");
            for (int d = 0; d < 24 && depth_end_us_[d].load() > 0; ++d) fprintf(stderr, " %d:%.1f", d, depth_end_us_[d].load() * 1e-3);
            fprintf(stderr, "; all leaves ordered at %.1f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start_).count());
        }
        return order_;
    }

private:
    const Graph& g_;
    int leaf_;
    int ml_min_ = 64;      // below this size only BFS level-set separators are tried (256..4000 measured: faster ordering, but more fill and slower sweeps)
    // recursion levels whose two halves run on their own threads (halves below 2000 vertices stay on their parent's
    // thread).  The subtrees are unequal, so more threads than usable CPUs still shorten the critical path: on a 16-CPU
    // quota the C3 analysis took 0.56 / 0.47 / 0.45 / 0.42 s with 3 / 4 / 5 / 6 levels.
    int par_depth_ = host_cpu_budget() >= 16 ? 6 : host_cpu_budget() >= 8 ? 5 : host_cpu_budget() >= 4 ? 4 : 3;
    std::vector<int> part_, vis_, lvl_, order_, loc_;
    std::atomic<int> next_id_{0}, stamp_{0};          // sub-problems run on disjoint vertex sets, possibly in parallel
    std::chrono::steady_clock::time_point t_start_;
    std::atomic<long long> depth_end_us_[24] = {};    // (trace) when the last separator of each recursion depth was found
    struct Bfs { std::vector<int> queue, lptr; int stamp = 0; };

    struct Split {
        std::vector<int> a, b, s;
        double cost = 1e300;
    };

    // multilevel (edge-cut minimising) separator of the induced subgraph; replaces `best` if smaller
    void multilevel_candidate(const std::vector<int>& verts, int id, uint64_t seed, Split& best) {
        int m = (int)verts.size();
        WGraph w;
        w.n = m;
        w.xadj.assign(m + 1, 0);
        w.vwgt.assign(m, 1);
        for (int i = 0; i < m; ++i) loc_[verts[i]] = i;
        {   // induced subgraph: degrees, offsets, lists -- ranges of vertices on host threads for the large subgraphs at the top
            const int nparts = m >= ND_PAR_MIN_N ? ND_NPART : 1;
            run_parts(nparts, [&](int pt) {
                const int i0 = (int)((int64_t)m * pt / nparts), i1 = (int)((int64_t)m * (pt + 1) / nparts);
                for (int i = i0; i < i1; ++i) {
                    const int v = verts[i];
                    int d = 0;
                    for (int64_t t = g_.xadj[v]; t < g_.xadj[v + 1]; ++t) d += part_[g_.adj[t]] == id;
                    w.xadj[i + 1] = d;
                }
            });
            for (int i = 0; i < m; ++i) w.xadj[i + 1] += w.xadj[i];
            w.adj.resize(w.xadj[m]);
            w.ewgt.resize(w.xadj[m]);
            run_parts(nparts, [&](int pt) {
                const int i0 = (int)((int64_t)m * pt / nparts), i1 = (int)((int64_t)m * (pt + 1) / nparts);
                for (int i = i0; i < i1; ++i) {
                    const int v = verts[i];
                    int o = w.xadj[i];
                    for (int64_t t = g_.xadj[v]; t < g_.xadj[v + 1]; ++t) if (part_[g_.adj[t]] == id) { w.adj[o] = loc_[g_.adj[t]]; w.ewgt[o] = 1; ++o; }
                }
            });
        }
        static const int trials2_min = getenv("FEASTHIP_ND_TRIALS2_MIN") ? atoi(getenv("FEASTHIP_ND_TRIALS2_MIN")) : 20000;
        int trials = m >= trials2_min ? 2 : 1;     // the top separators dominate the fill: try harder there
        // the trials are independent (fixed seeds): run them side by side, then judge them in trial order so that the
        // outcome does not depend on thread timing
        std::vector<std::vector<int>> labels(trials);
        std::vector<char> okv(trials, 0);
        {
            std::vector<std::thread> pool;
            for (int tr = 1; tr < trials; ++tr)
                pool.emplace_back([&, tr]() { okv[tr] = multilevel_separator(w, labels[tr], seed * 7919u + 17u + 104729u * tr) ? 1 : 0; });
            okv[0] = multilevel_separator(w, labels[0], seed * 7919u + 17u) ? 1 : 0;
            for (auto& th : pool) th.join();
        }
        for (int tr = 0; tr < trials; ++tr) {
            if (!okv[tr]) continue;
            const std::vector<int>& label = labels[tr];
            Split sp;
            for (int i = 0; i < m; ++i) (label[i] == 0 ? sp.a : label[i] == 1 ? sp.b : sp.s).push_back(verts[i]);
            if (sp.a.empty() || sp.b.empty() || sp.s.empty()) continue;
            double bal = (double)std::max(sp.a.size(), sp.b.size()) / (double)(m - (int)sp.s.size());
            if (bal > 0.68) continue;
            // separator size first; among separators of (nearly) the same size the better balanced one (as for the level-set candidates)
            // (8 seeds on C3's graph: mean factor flops 4.997e11 with 2 trials, 5.058e11 with 3 or 4 -- a third trial buys balance with fill)
            sp.cost = (double)sp.s.size() + 1e-3 * std::abs((double)sp.a.size() - (double)sp.b.size());
            if (sp.cost < best.cost) best = std::move(sp);
        }
    }

    // BFS restricted to part_ == id.  Fills queue_ (visit order) and lptr_ (level offsets).
    void bfs(int root, int id, Bfs& B) {
        std::vector<int>& queue_ = B.queue;
        std::vector<int>& lptr_ = B.lptr;
        const int stamp_ = B.stamp = ++this->stamp_;
        queue_.clear();
        lptr_.clear();
        queue_.push_back(root);
        vis_[root] = stamp_;
        lvl_[root] = 0;
        lptr_.push_back(0);
        size_t head = 0;
        int cur = 0;
        while (head < queue_.size()) {
            int v = queue_[head];
            if (lvl_[v] != cur) { cur = lvl_[v]; lptr_.push_back((int)head); }
            ++head;
            for (int64_t t = g_.xadj[v]; t < g_.xadj[v + 1]; ++t) {
                int w = g_.adj[t];
                if (part_[w] == id && vis_[w] != stamp_) {
                    vis_[w] = stamp_;
                    lvl_[w] = lvl_[v] + 1;
                    queue_.push_back(w);
                }
            }
        }
        lptr_.push_back((int)queue_.size());
    }

    // Evaluate the best level-set separator of the current BFS structure.
    void candidate(int m, Split& best, const Bfs& B) {
        const std::vector<int>& queue_ = B.queue;
        const std::vector<int>& lptr_ = B.lptr;
        const int stamp_ = B.stamp;
        int nl = (int)lptr_.size() - 1;
        if (nl < 3) return;
        int bj = -1;
        double bcost = 1e300;
        // only levels that leave both sides with at least 35% of the vertices; among them the smallest
        for (int j = 1; j < nl - 1; ++j) {
            int below = lptr_[j], sz = lptr_[j + 1] - lptr_[j], above = m - below - sz;
            if (std::min(below, above) < 0.35 * (m - sz)) continue;
            double cost = sz + 1e-3 * std::abs(below - above);
            if (cost < bcost) { bcost = cost; bj = j; }
        }
        if (bj < 0) {   // no balanced level: take the one closest to the median
            int bestd = INT32_MAX;
            for (int j = 1; j < nl - 1; ++j) {
                int below = lptr_[j], sz = lptr_[j + 1] - lptr_[j], above = m - below - sz;
                if (below <= 0 || above <= 0) continue;
                int d = std::abs(below - above);
                if (d < bestd) { bestd = d; bj = j; bcost = sz * (1.0 + 4.0 * (double)d / m); }
            }
        }
        if (bj < 0 || bcost >= best.cost) return;
        // build trimmed split: vertices of level bj with no neighbour in level bj+1 join side a
        Split sp;
        sp.cost = 0;
        for (int k = 0; k < lptr_[bj]; ++k) sp.a.push_back(queue_[k]);
        for (int k = lptr_[bj + 1]; k < m; ++k) sp.b.push_back(queue_[k]);
        for (int k = lptr_[bj]; k < lptr_[bj + 1]; ++k) {
            int v = queue_[k];
            bool touches = false;
            for (int64_t t = g_.xadj[v]; t < g_.xadj[v + 1] && !touches; ++t) {
                int w = g_.adj[t];
                if (vis_[w] == stamp_ && lvl_[w] == bj + 1) touches = true;
            }
            if (touches) sp.s.push_back(v); else sp.a.push_back(v);
        }
        sp.cost = bcost;
        if (sp.cost < best.cost) best = std::move(sp);
    }

    void emit(const std::vector<int>& verts, int lo) {
        for (size_t k = 0; k < verts.size(); ++k) order_[lo + k] = verts[k];
    }

    void dissect(std::vector<int>& verts, int lo, int depth = 0) {
        int m = (int)verts.size();
        if (m <= leaf_) { emit(verts, lo); return; }
        int id = next_id_++;
        for (int v : verts) part_[v] = id;
        Bfs B;
        std::vector<int>& queue_ = B.queue;
        std::vector<int>& lptr_ = B.lptr;
        // large subgraphs: the multilevel candidate is computed on its own thread while this one walks the BFS level structures
        // (disjoint scratch: loc_ vs vis_/lvl_), the first of which also tells whether the subgraph is connected -- if it is not, the
        // candidate is dropped; merged below in the same order as the serial code
        Split best, best_ml;
        static const uint64_t seed_env = getenv("FEASTHIP_ND_SEED") ? (uint64_t)atoll(getenv("FEASTHIP_ND_SEED")) : 0;      // (experiments: the spread of the ordering quality over seeds)
        const uint64_t ml_seed = (uint64_t)lo * 2654435761ull + (uint64_t)m + seed_env * 0x9E3779B97F4A7C15ull;      // independent of thread timing
        std::thread tml;
        const bool ml_async = m >= 20000;      // (down to 1000 measured: no faster)
        if (ml_async) tml = std::thread([&]() { multilevel_candidate(verts, id, ml_seed, best_ml); });
        bfs(verts[0], id, B);
        if ((int)queue_.size() < m) {   // disconnected: order components independently
            if (ml_async) tml.join();
            std::vector<int> comp(queue_), rest;
            int st = B.stamp;
            for (int v : verts) if (vis_[v] != st) rest.push_back(v);
            verts.clear(); verts.shrink_to_fit();
            int ncomp = (int)comp.size();
            dissect(comp, lo, depth);
            dissect(rest, lo + ncomp, depth);
            return;
        }
        const bool tr_ = depth <= 2 && getenv("FEASTHIP_ANALYZE_TRACE");
        auto t_a = std::chrono::steady_clock::now();
        int height = (int)lptr_.size() - 1;
        candidate(m, best, B);
        for (int it = 0; it < 6; ++it) {
            // min-degree vertex of the last level
            int nl = (int)lptr_.size() - 1, r = -1;
            int64_t bd = INT64_MAX;
            for (int k = lptr_[nl - 1]; k < lptr_[nl]; ++k) {
                int v = queue_[k];
                int64_t d = g_.xadj[v + 1] - g_.xadj[v];
                if (d < bd) { bd = d; r = v; }
            }
            bfs(r, id, B);
            candidate(m, best, B);
            int h2 = (int)lptr_.size() - 1;
            if (h2 <= height) break;
            height = h2;
        }
        auto t_b = std::chrono::steady_clock::now();
        if (ml_async) { tml.join(); if (best_ml.cost < best.cost) best = std::move(best_ml); }
        else if (m >= ml_min_) multilevel_candidate(verts, id, ml_seed, best);
        // (until late in round 3 this `else` hung on the trace line below: without FEASTHIP_ANALYZE_TRACE the three top levels computed
        // their multilevel separator a second time, serially, with the same seed and the same result -- 57 ms of every untraced analysis)
        if (tr_) fprintf(stderr, "[feasthip nd depth %d m=%d] bfs candidates %.1f ms, + wait for multilevel %.1f ms\n", depth, m,
                         std::chrono::duration<double, std::milli>(t_b - t_a).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_b).count());
        if (depth < 24) {
            const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_start_).count();
            long long cur = depth_end_us_[depth].load();
            while (us > cur && !depth_end_us_[depth].compare_exchange_weak(cur, us)) {}
        }
        if (best.s.empty() || best.a.empty() || best.b.empty()) { emit(verts, lo); return; }
        verts.clear(); verts.shrink_to_fit();
        int na = (int)best.a.size(), nb = (int)best.b.size();
        emit(best.s, lo + na + nb);
        std::vector<int> a(std::move(best.a)), b(std::move(best.b));
        best.s.clear(); best.s.shrink_to_fit();
        if (depth < par_depth_ && std::min(na, nb) >= 2000) {     // the two halves are independent: 16 (or 32) host threads
            std::thread th([&]() { dissect(a, lo, depth + 1); });
            dissect(b, lo + na, depth + 1);
            th.join();
        } else {
            dissect(a, lo, depth + 1);
            dissect(b, lo + na, depth + 1);
        }
    }
};

void finish_layout(Symbolic& S) {
    int ns = S.nsuper;
    S.depth.assign(ns, 0);
    S.maxdepth = 0;
    for (int s = ns - 1; s >= 0; --s) {
        S.depth[s] = S.sn_parent[s] < 0 ? 0 : S.depth[S.sn_parent[s]] + 1;
        S.maxdepth = std::max(S.maxdepth, S.depth[s]);
    }
    S.levels.assign(S.maxdepth + 1, {});
    for (int s = 0; s < ns; ++s) S.levels[S.depth[s]].push_back(s);
    S.lp_off.assign(ns, 0);
    S.f12_off.assign(ns, 0);
    S.cb_off.assign(ns, 0);
    S.crow_off.assign(ns, 0);
    S.cb_size[0] = S.cb_size[1] = 0;
    S.crow_size[0] = S.crow_size[1] = 0;
    int64_t off = 0;
    S.nnz_l = S.nnz_u = 0;
    S.flops_factor = 0;
    S.max_p = S.max_front = 0;
    for (int s = 0; s < ns; ++s) {
        int64_t p = S.p(s), u = S.u(s);
        S.lp_off[s] = off; off += (p + u) * p;
        S.f12_off[s] = off; off += p * u;
        S.nnz_l += p * (p + 1) / 2 + u * p;
        S.nnz_u += p * (p - 1) / 2 + u * p;
        double m = (double)(p + u);
        for (int64_t k = 0; k < p; ++k) { double r = m - k - 1; S.flops_factor += 8.0 * (r * r) + 6.0 * r; }
        S.max_p = std::max<int>(S.max_p, (int)p);
        S.max_front = std::max<int>(S.max_front, (int)(p + u));
    }
    S.fact_size = off;
    std::vector<int64_t> cbl(S.maxdepth + 1, 0), crl(S.maxdepth + 1, 0);
    for (int s = 0; s < ns; ++s) {
        int d = S.depth[s];
        int64_t u = S.u(s);
        S.cb_off[s] = cbl[d]; cbl[d] += u * u;
        S.crow_off[s] = crl[d]; crl[d] += u;
    }
    S.cb_used = cbl;
    S.crow_used = crl;
    S.sum_u = 0;
    for (int s = 0; s < ns; ++s) S.sum_u += S.u(s);
    for (int d = 0; d <= S.maxdepth; ++d) {
        S.cb_size[d & 1] = std::max(S.cb_size[d & 1], cbl[d]);
        S.crow_size[d & 1] = std::max(S.crow_size[d & 1], crl[d]);
    }
}

}  // namespace

void analyze_dense(int n, Symbolic& S) {
    S = Symbolic();
    S.n = n;
    S.dense = true;
    S.perm.resize(n);
    std::iota(S.perm.begin(), S.perm.end(), 0);
    S.iperm = S.perm;
    S.nsuper = 1;
    S.sn_start = {0, n};
    S.sn_of.assign(n, 0);
    S.sn_parent = {-1};
    S.struct_ptr = {0, 0};
    S.child_ptr = {0, 0};
    finish_layout(S);
}

std::string analyze_pattern(int n, const int64_t* colptr, const int64_t* rowidx, const SymbolicOptions& opt, Symbolic& S) {
    S = Symbolic();
    S.n = n;
    if (n <= 0) return "empty matrix";
    for (int j = 0; j < n; ++j) {
        if (colptr[j + 1] < colptr[j]) return "column pointers not monotone";
        for (int64_t t = colptr[j]; t < colptr[j + 1]; ++t)
            if (rowidx[t] < 0 || rowidx[t] >= n) return "row index out of range";
    }
    const bool trace = getenv("FEASTHIP_ANALYZE_TRACE") != nullptr;
    auto tph = std::chrono::steady_clock::now();
    Graph g = symmetrize(n, colptr, rowidx);

    auto phase = [&](const char* name) {
        if (!trace) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[feasthip analyze] %-28s %.1f ms\n", name, std::chrono::duration<double, std::milli>(t1 - tph).count());
        tph = t1;
    };
    phase("A + A^T");
    auto sub = [&](const char* name) { phase(name); };
    // ---- 1. fill-reducing ordering --------------------------------------
    std::vector<int> perm(n), iperm(n);
    if (opt.ordering == 1) {
        std::iota(perm.begin(), perm.end(), 0);
    } else {
        NestedDissection nd(g, std::max(1, opt.leaf_size));
        perm = nd.run();
    }
    for (int k = 0; k < n; ++k) {
        if (perm[k] < 0 || perm[k] >= n) return "internal: ordering incomplete";
        iperm[perm[k]] = k;
    }
    phase("ordering");
    Graph h;
    permute_graph(g, perm, iperm, h);
    sub("  permuted graph");
    // ---- 2. elimination tree (Liu) + postorder ---------------------------
    std::vector<int> parent(n, -1), anc(n, -1);
    for (int j = 0; j < n; ++j)
        for (int64_t t = h.xadj[j]; t < h.xadj[j + 1]; ++t) {
            int i = h.adj[t];
            if (i >= j) continue;
            int r = i;
            while (anc[r] != -1 && anc[r] != j) { int nx = anc[r]; anc[r] = j; r = nx; }
            if (anc[r] == -1) { anc[r] = j; parent[r] = j; }
        }
    sub("  etree");
    {   // postorder (iterative DFS, children visited in increasing index)
        std::vector<int> head(n, -1), next(n, -1), post;
        post.reserve(n);
        for (int j = n - 1; j >= 0; --j)
            if (parent[j] >= 0) { next[j] = head[parent[j]]; head[parent[j]] = j; }
        std::vector<int> stack;
        for (int r = 0; r < n; ++r) {
            if (parent[r] >= 0) continue;
            stack.push_back(r);
            while (!stack.empty()) {
                int v = stack.back();
                int c = head[v];
                if (c >= 0) { head[v] = next[c]; stack.push_back(c); }
                else { post.push_back(v); stack.pop_back(); }
            }
        }
        std::vector<int> inv(n), perm2(n), parent2(n);
        for (int k = 0; k < n; ++k) inv[post[k]] = k;
        for (int k = 0; k < n; ++k) {
            perm2[k] = perm[post[k]];
            parent2[k] = parent[post[k]] < 0 ? -1 : inv[parent[post[k]]];
        }
        perm.swap(perm2);
        parent.swap(parent2);
        for (int k = 0; k < n; ++k) iperm[perm[k]] = k;
        sub("  postorder");
        permute_graph(g, perm, iperm, h);
    }

    phase("  permuted graph 2");
    // ---- 3. column counts (Gilbert, Ng, Peyton: skeleton leaves + least common ancestors, O(nnz(A) alpha)) ----
    // cc[j] = number of off-diagonal entries of column j of L.  The graph is postordered, so post(k) = k.
    std::vector<int> cc(n, 0), nchild(n, 0);
    {
        std::vector<int> first(n, -1), maxfirst(n, -1), prevleaf(n, -1), ancestor(n), delta(n, 0);
        for (int k = 0; k < n; ++k) {
            int j = k;
            delta[j] = first[j] == -1 ? 1 : 0;                 // j is a leaf of the elimination tree
            for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = k;
        }
        for (int i = 0; i < n; ++i) ancestor[i] = i;
        for (int j = 0; j < n; ++j) {
            if (parent[j] != -1) delta[parent[j]]--;
            for (int64_t t = h.xadj[j]; t < h.xadj[j + 1]; ++t) {
                int i = h.adj[t];
                if (i <= j || first[j] <= maxfirst[i]) continue;    // j is not a leaf of the i-th row subtree
                maxfirst[i] = first[j];
                int jprev = prevleaf[i];
                prevleaf[i] = j;
                delta[j]++;
                if (jprev != -1) {                                  // subsequent leaf: remove the overlap at the least common ancestor
                    int q = jprev;
                    while (q != ancestor[q]) q = ancestor[q];
                    for (int sv = jprev; sv != q;) { int sp = ancestor[sv]; ancestor[sv] = q; sv = sp; }
                    delta[q]--;
                }
            }
            if (parent[j] != -1) ancestor[j] = parent[j];
        }
        for (int j = 0; j < n; ++j) if (parent[j] != -1) delta[parent[j]] += delta[j];
        for (int j = 0; j < n; ++j) cc[j] = delta[j] - 1;           // without the diagonal
    }
    for (int j = 0; j < n; ++j) if (parent[j] >= 0) nchild[parent[j]]++;

    phase("column counts");
    // ---- 4. fundamental supernodes ----------------------------------------
    std::vector<int> fstart;   // first column of each fundamental supernode
    for (int j = 0; j < n; ++j) {
        bool join = j > 0 && parent[j - 1] == j && cc[j - 1] == cc[j] + 1 && nchild[j] == 1;
        if (!join) fstart.push_back(j);
    }
    int nf = (int)fstart.size();
    fstart.push_back(n);
    std::vector<int> fof(n);
    for (int s = 0; s < nf; ++s) for (int j = fstart[s]; j < fstart[s + 1]; ++j) fof[j] = s;
    std::vector<int> fparent(nf, -1);
    for (int s = 0; s < nf; ++s) {
        int last = fstart[s + 1] - 1;
        fparent[s] = parent[last] < 0 ? -1 : fof[parent[last]];
    }

    phase("fundamental supernodes");
    // ---- 5. relaxed amalgamation on the supernode tree --------------------
    std::vector<int64_t> P(nf), U(nf);
    std::vector<double> Z(nf, 0.0);
    std::vector<char> dead(nf, 0);
    // children and members (fundamental supernodes merged into a node) as singly linked lists in flat arrays with tail pointers: a merge
    // appends the child's lists in O(1) (vectors of vectors cost 60 k small allocations here: 3.7 of the analysis' milliseconds)
    std::vector<int> kid_head(nf, -1), kid_tail(nf, -1), kid_next(nf, -1), mem_head(nf), mem_tail(nf), mem_next(nf, -1);
    for (int s = 0; s < nf; ++s) {
        P[s] = fstart[s + 1] - fstart[s];
        U[s] = cc[fstart[s + 1] - 1];
        mem_head[s] = mem_tail[s] = s;
    }
    for (int s = 0; s < nf; ++s) {
        const int ps = fparent[s];
        if (ps < 0) continue;
        if (kid_tail[ps] < 0) kid_head[ps] = s; else kid_next[kid_tail[ps]] = s;
        kid_tail[ps] = s;
    }
    for (int s = 0; s < nf; ++s) {   // children have smaller indices (postorder) and are final by now
        for (;;) {
            // candidate with the fewest new zeros first (the first of them in list order)
            int bc = -1, bprev = -1;
            double bz = 1e300;
            for (int c = kid_head[s], prev = -1; c >= 0; prev = c, c = kid_next[c]) {
                double nz = (double)P[c] * (double)(P[s] + U[s] - U[c]);
                if (nz < bz) { bz = nz; bc = c; bprev = prev; }
            }
            if (bc < 0) break;
            const int c = bc;
            int64_t pn = P[s] + P[c];
            double zn = Z[s] + Z[c] + bz;
            double tot = (double)pn * (pn + 1) / 2.0 + (double)pn * (double)U[s];
            double zf = zn / tot;
            bool ok = pn <= opt.relax_small || (pn <= opt.relax_mid && zf <= opt.zmid) || zf <= opt.zbig;
            if (!ok) break;
            // merge c into s: c leaves the list of children, its own children and members are appended
            P[s] = pn;
            Z[s] = zn;
            dead[c] = 1;
            if (bprev < 0) kid_head[s] = kid_next[c]; else kid_next[bprev] = kid_next[c];
            if (kid_tail[s] == c) kid_tail[s] = bprev;
            kid_next[c] = -1;
            if (kid_head[c] >= 0) {
                if (kid_tail[s] < 0) kid_head[s] = kid_head[c]; else kid_next[kid_tail[s]] = kid_head[c];
                kid_tail[s] = kid_tail[c];
                kid_head[c] = kid_tail[c] = -1;
            }
            mem_next[mem_tail[s]] = mem_head[c];
            mem_tail[s] = mem_tail[c];
        }
    }
    sub("  amalgamation merges");
    // final numbering: DFS over the amalgamated tree, children first
    std::vector<int> sn_parent_new;         // parent of every final supernode (known from the amalgamated tree before any row structure is)
    {
        std::vector<int> newpos;   // newpos[k] = old (postordered) column placed at k
        newpos.reserve(n);
        std::vector<int> sstart, newid(nf, -1), cols;
        std::vector<std::pair<int, int>> stack;      // (node, next child to visit)
        for (int r = 0; r < nf; ++r) {
            if (dead[r] || fparent[r] >= 0) continue;
            // a live root: note fparent of merged nodes is irrelevant, the child lists hold the tree
            stack.push_back({r, kid_head[r]});
            while (!stack.empty()) {
                auto& top = stack.back();
                int v = top.first;
                if (top.second >= 0) { int c = top.second; top.second = kid_next[c]; stack.push_back({c, kid_head[c]}); }
                else {
                    cols.clear();
                    for (int f = mem_head[v]; f >= 0; f = mem_next[f]) for (int j = fstart[f]; j < fstart[f + 1]; ++j) cols.push_back(j);
                    std::sort(cols.begin(), cols.end());
                    newid[v] = (int)sstart.size();
                    sstart.push_back((int)newpos.size());
                    newpos.insert(newpos.end(), cols.begin(), cols.end());
                    stack.pop_back();
                }
            }
        }
        if ((int)newpos.size() != n) return "internal: amalgamation lost columns";
        sstart.push_back(n);
        sn_parent_new.assign(sstart.size() - 1, -1);
        for (int v = 0; v < nf; ++v)
            if (newid[v] >= 0) for (int c = kid_head[v]; c >= 0; c = kid_next[c]) sn_parent_new[newid[c]] = newid[v];
        std::vector<int> perm2(n);
        for (int k = 0; k < n; ++k) perm2[k] = perm[newpos[k]];
        perm.swap(perm2);
        for (int k = 0; k < n; ++k) iperm[perm[k]] = k;
        S.sn_start = sstart;
        S.nsuper = (int)sstart.size() - 1;
    }
    sub("  final numbering");
    permute_graph(g, perm, iperm, h);
    S.perm = perm;
    S.iperm = iperm;
    int ns = S.nsuper;
    S.sn_of.resize(n);
    for (int s = 0; s < ns; ++s) for (int j = S.sn_start[s]; j < S.sn_start[s + 1]; ++j) S.sn_of[j] = s;

    phase("  permuted graph 3");
    // ---- 6. supernode row structures, tree, relative indices ---------------
    S.sn_parent.assign(ns, -1);
    S.struct_ptr.assign(ns + 1, 0);
    std::vector<std::vector<int>> ch(ns);
    std::vector<std::vector<int>> st(ns);
    // The tree is known already (sn_parent_new, from the amalgamation), children before parents, every subtree a contiguous range of
    // supernodes: disjoint subtrees are independent (a structure depends on the columns' adjacency and on the children's structures
    // only), so the subtrees below a weight threshold go to the host threads and the few supernodes above them follow serially.
    for (int s2 = 0; s2 < ns; ++s2) if (sn_parent_new[s2] >= 0) ch[sn_parent_new[s2]].push_back(s2);
    std::atomic<int> bad_tree{0};
    auto structure_of = [&](int s2, std::vector<int>& stamp, std::vector<int>& rows) {
        int last = S.sn_start[s2 + 1] - 1;
        rows.clear();
        for (int j = S.sn_start[s2]; j <= last; ++j)
            for (int64_t t = h.xadj[j]; t < h.xadj[j + 1]; ++t) {
                int i = h.adj[t];
                if (i > last && stamp[i] != s2) { stamp[i] = s2; rows.push_back(i); }
            }
        for (int c : ch[s2])
            for (int i : st[c])
                if (i > last && stamp[i] != s2) { stamp[i] = s2; rows.push_back(i); }
        std::sort(rows.begin(), rows.end());
        st[s2] = rows;
        const int ps = rows.empty() ? -1 : S.sn_of[rows[0]];
        if (ps != sn_parent_new[s2]) bad_tree.store(1);
        S.sn_parent[s2] = ps;
    };
    {
        std::vector<int64_t> wsub(ns, 0);
        int64_t wtot = 0;
        for (int s2 = 0; s2 < ns; ++s2) {
            wsub[s2] += h.xadj[S.sn_start[s2 + 1]] - h.xadj[S.sn_start[s2]] + 8;
            wtot += h.xadj[S.sn_start[s2 + 1]] - h.xadj[S.sn_start[s2]] + 8;
            if (sn_parent_new[s2] >= 0) wsub[sn_parent_new[s2]] += wsub[s2];
        }
        const int64_t wmax = n >= 20000 ? wtot / 48 : wtot + 1;      // small problems: everything "above" the (empty) set of tasks
        std::vector<int> first_desc(ns);
        for (int s2 = 0; s2 < ns; ++s2) first_desc[s2] = ch[s2].empty() ? s2 : first_desc[ch[s2][0]];
        std::vector<std::pair<int, int>> tasks;       // [first, last] supernode ranges of the independent subtrees
        std::vector<char> above(ns, 0);
        for (int s2 = 0; s2 < ns; ++s2) {
            if (wsub[s2] > wmax) { above[s2] = 1; continue; }
            const int ps = sn_parent_new[s2];
            if (ps < 0 || wsub[ps] > wmax) tasks.push_back({first_desc[s2], s2});
        }
        std::sort(tasks.begin(), tasks.end(), [&](const std::pair<int, int>& a, const std::pair<int, int>& b) { return wsub[a.second] != wsub[b.second] ? wsub[a.second] > wsub[b.second] : a.second < b.second; });
        std::atomic<int> next_task{0};
        const int ntasks = (int)tasks.size();
        if (ntasks > 1)
            run_parts(std::min(ntasks, 16), [&](int) {
                std::vector<int> stamp, rows;
                for (;;) {
                    const int t = next_task.fetch_add(1);
                    if (t >= ntasks) break;
                    if (stamp.empty()) stamp.assign(n, -1);
                    for (int s2 = tasks[t].first; s2 <= tasks[t].second; ++s2) structure_of(s2, stamp, rows);
                }
            });
        std::vector<int> stamp(n, -1), rows;
        if (ntasks <= 1) { for (int s2 = 0; s2 < ns; ++s2) structure_of(s2, stamp, rows); }
        else for (int s2 = 0; s2 < ns; ++s2) if (above[s2]) structure_of(s2, stamp, rows);
    }
    if (bad_tree.load()) return "internal: amalgamated tree and row structures disagree";
    for (int s2 = 0; s2 < ns; ++s2) S.struct_ptr[s2 + 1] = S.struct_ptr[s2] + (int64_t)st[s2].size();
    sub("  row structures");
    S.struct_idx.resize(S.struct_ptr[ns]);
    S.rel.resize(S.struct_ptr[ns]);
    S.child_ptr.assign(ns + 1, 0);
    for (int s = 0; s < ns; ++s) {
        std::copy(st[s].begin(), st[s].end(), S.struct_idx.begin() + S.struct_ptr[s]);
        S.child_ptr[s + 1] = S.child_ptr[s] + (int64_t)ch[s].size();
        S.child_idx.insert(S.child_idx.end(), ch[s].begin(), ch[s].end());
    }
    for (int c = 0; c < ns; ++c) {
        int ps = S.sn_parent[c];
        if (ps < 0) continue;
        int c0 = S.sn_start[ps], pp = S.sn_start[ps + 1] - c0;
        const std::vector<int>& pst = st[ps];
        size_t q = 0;
        for (size_t k = 0; k < st[c].size(); ++k) {
            int i = st[c][k];
            int r;
            if (i < c0 + pp) r = i - c0;
            else {
                while (q < pst.size() && pst[q] < i) ++q;
                if (q >= pst.size() || pst[q] != i) return "internal: child structure not contained in parent front";
                r = pp + (int)q;
            }
            S.rel[S.struct_ptr[c] + k] = r;
        }
    }
    phase("  relative indices");
    finish_layout(S);
    phase("layout");
    return "";
}

int64_t entry_dest(const Symbolic& S, int i_old, int j_old) {
    int i = S.iperm[i_old], j = S.iperm[j_old];
    int m = std::min(i, j);
    int s = S.sn_of[m];
    int c0 = S.sn_start[s];
    int64_t p = S.p(s), u = S.u(s);
    const int* st = S.struct_idx.data() + S.struct_ptr[s];
    auto pos = [&](int r) -> int64_t {
        const int* it = std::lower_bound(st, st + u, r);
        if (it == st + u || *it != r) return -1;
        return (int64_t)(it - st);
    };
    if (i >= j) {   // lower triangle incl. diagonal -> L panel
        int64_t col = j - c0, row;
        if (i < c0 + p) row = i - c0;
        else { int64_t q = pos(i); if (q < 0) return -1; row = p + q; }
        return S.lp_off[s] + row + col * (p + u);
    }
    int64_t row = i - c0;
    if (j < c0 + p) return S.lp_off[s] + row + (int64_t)(j - c0) * (p + u);
    int64_t q = pos(j);
    if (q < 0) return -1;
    return S.f12_off[s] + row + q * p;
}

}  // namespace feasthip
