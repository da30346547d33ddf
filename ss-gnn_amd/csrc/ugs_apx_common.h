// ugs_apx_common.h -- graph and APX-DD order of the apx_ugs_sampler entry points (host restatement ugs_apx.cpp, GPU variant
// ugs_apx_gpu.hip).  Reference: src/samplers/apx_ugs_sampler/src/apx_ugs_sampler.cpp:15-168.  The order is a host computation
// in both variants; only the generator behind `rs.below(n)` differs (the reference's single mt19937_64 stream there, a
// counter-based generator here).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <queue>
#include <unordered_set>
#include <vector>

namespace ugs_apx {

struct SimpleGraph {                  // sorted, duplicate-free adjacency of the first graph
    int n = 0;
    std::vector<int64_t> off;
    std::vector<int> nbr;
    int deg(int v) const { return (int)(off[(size_t)v + 1] - off[(size_t)v]); }
    const int *row(int v) const { return nbr.data() + off[(size_t)v]; }
    bool adjacent(int u, int v) const {
        if (deg(u) < deg(v)) std::swap(u, v);
        return std::binary_search(row(v), row(v) + deg(v), u);
    }
};

inline SimpleGraph read_graph(const int64_t *src, const int64_t *dst, int64_t c0, int64_t c1) {
    SimpleGraph g;
    for (int64_t j = c0; j < c1; ++j) {
        g.n = std::max(g.n, (int)src[j] + 1);
        g.n = std::max(g.n, (int)dst[j] + 1);
    }
    std::vector<std::vector<int>> lists((size_t)g.n);
    for (int64_t j = c0; j < c1; ++j) {
        const int u = (int)src[j], v = (int)dst[j];
        lists[(size_t)u].push_back(v);
        lists[(size_t)v].push_back(u);
    }
    g.off.assign((size_t)g.n + 1, 0);
    for (int v = 0; v < g.n; ++v) {
        auto &l = lists[(size_t)v];
        std::sort(l.begin(), l.end());
        l.erase(std::unique(l.begin(), l.end()), l.end());
        g.off[(size_t)v + 1] = g.off[(size_t)v] + (int64_t)l.size();
    }
    g.nbr.reserve((size_t)g.off[(size_t)g.n]);
    for (int v = 0; v < g.n; ++v) g.nbr.insert(g.nbr.end(), lists[(size_t)v].begin(), lists[(size_t)v].end());
    return g;
}

struct Order { std::vector<int> seq, pos; std::vector<double> est; };

// APX-DD: degree order refined by sampled "later neighbour" fractions; bucket estimates deg^5
template <class RS>
Order dominating_order(const SimpleGraph &g, int k, double beta, RS &rs) {
    const int n = g.n;
    const double eta = std::pow(beta, 1.0 / static_cast<double>(k - 1)) / (6.0 * k * k);
    const int h = static_cast<int>(std::ceil(10.0 / (eta * eta) * std::log(n)));
    Order o;
    o.seq.resize((size_t)n); o.pos.resize((size_t)n); o.est.assign((size_t)n, 0.0);
    std::vector<double> score((size_t)n);
    for (int v = 0; v < n; ++v) { score[(size_t)v] = static_cast<double>(g.deg(v)); o.seq[(size_t)v] = v; }
    auto by_score = [&](int a, int b) { return score[(size_t)a] != score[(size_t)b] ? score[(size_t)a] > score[(size_t)b] : a > b; };
    std::sort(o.seq.begin(), o.seq.end(), by_score);
    for (int i = 0; i < n; ++i) o.pos[(size_t)o.seq[(size_t)i]] = i;
    for (int idx = 0; idx < n; ++idx) {
        const int v = o.seq[(size_t)idx];
        const int d = g.deg(v);
        if (d == 0) { o.est[(size_t)v] = 0.0; continue; }
        int later = 0;
        for (int i = 0; i < h; ++i) {
            const int u = g.row(v)[rs.below(d)];
            if (o.pos[(size_t)v] < o.pos[(size_t)u]) ++later;
        }
        if (later >= 2.0 * eta * h) {
            o.est[(size_t)v] = std::pow(static_cast<double>(d), 5.0);
        } else {
            o.est[(size_t)v] = 0.0;
            score[(size_t)v] = 3.0 * eta * d;
            std::sort(o.seq.begin() + idx, o.seq.end(), by_score);
            for (int i = idx; i < n; ++i) o.pos[(size_t)o.seq[(size_t)i]] = i;
        }
    }
    const double small = static_cast<double>(k) / eta;
    for (int v = 0; v < n; ++v) {
        if (g.deg(v) > small) continue;
        std::queue<int> q;
        std::unordered_set<int> seen;
        q.push(v); seen.insert(v);
        bool enough = false;
        while (!q.empty() && seen.size() < static_cast<size_t>(k)) {
            const int u = q.front(); q.pop();
            for (int t = 0; t < g.deg(u); ++t) {
                const int w = g.row(u)[t];
                if (seen.find(w) == seen.end() && o.pos[(size_t)v] < o.pos[(size_t)w]) {
                    seen.insert(w); q.push(w);
                    if (seen.size() >= static_cast<size_t>(k)) { enough = true; break; }
                }
            }
        }
        if (enough) {
            int inside = 0;
            for (int t = 0; t < g.deg(v); ++t) if (o.pos[(size_t)v] < o.pos[(size_t)g.row(v)[t]]) ++inside;
            o.est[(size_t)v] = std::pow(static_cast<double>(inside), 5.0);
        } else {
            o.est[(size_t)v] = 0.0;
        }
    }
    return o;
}

}  // namespace ugs_apx
