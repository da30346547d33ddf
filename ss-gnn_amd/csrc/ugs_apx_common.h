// ugs_apx_common.h -- what the two apx_ugs_sampler backends share (sequential host backend ugs_apx.cpp, GPU variant ugs_apx_gpu.hip):
// the graph of the first batch entry as a flat CSR, the vertex ranking with its bucket weights, and the sampling budget of a cut
// estimate.  Behavioural contract: reference src/samplers/apx_ugs_sampler/src/apx_ugs_sampler.cpp:15-168 (graph, APX-DD order) and
// :184-205 (budget).  What is contractual is WHICH random draws are made, in which sequence, and the floating-point expressions
// that decide comparisons; the data structures here are this repo's own: one counting pass builds the CSR, adjacency tests go
// through a bit matrix, the ranking is kept sorted by moving ONE vertex (only the processed vertex's score ever changes), and the
// reachability test is a bounded flood with stamp marks and a k-entry ring.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace ugs_apx {

// ---------------------------------------------------------------------------------------------------------------------
// Graph: sorted duplicate-free rows (the reference symmetrises every column and removes repeats, :27-39; a self loop stays as an
// entry of its own row).  Columns with a negative endpoint are ignored (undefined in the reference).
// ---------------------------------------------------------------------------------------------------------------------
struct Csr {
    int n = 0;
    std::vector<int64_t> off;          // [n + 1]
    std::vector<int> nbr;              // rows ascending, no repeats
    std::vector<uint64_t> bits;        // n x words bit matrix when the graph is small enough, else empty
    int words = 0;

    int deg(int v) const { return (int)(off[(size_t)v + 1] - off[(size_t)v]); }
    const int *row(int v) const { return nbr.data() + off[(size_t)v]; }
    bool linked(int a, int b) const {
        if (words) return (bits[(size_t)a * (size_t)words + ((size_t)b >> 6)] >> (b & 63)) & 1ull;
        if (deg(a) > deg(b)) { const int t = a; a = b; b = t; }       // search the shorter row
        const int *r = row(a);
        int lo = 0, hi = deg(a);
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (r[mid] < b) lo = mid + 1; else hi = mid; }
        return lo < deg(a) && r[lo] == b;
    }
};

inline Csr csr_of_columns(const int64_t *src, const int64_t *dst, int64_t c0, int64_t c1) {
    Csr g;
    int64_t top = -1;
    for (int64_t j = c0; j < c1; ++j) {
        if (src[j] < 0 || dst[j] < 0) continue;
        if (src[j] > top) top = src[j];
        if (dst[j] > top) top = dst[j];
    }
    g.n = (int)(top + 1);
    const size_t n = (size_t)g.n;
    // counting pass: every column is entered at both ends
    std::vector<int64_t> fill(n + 1, 0);
    for (int64_t j = c0; j < c1; ++j) {
        if (src[j] < 0 || dst[j] < 0) continue;
        ++fill[(size_t)src[j] + 1];
        ++fill[(size_t)dst[j] + 1];
    }
    for (size_t v = 0; v < n; ++v) fill[v + 1] += fill[v];
    std::vector<int> raw((size_t)fill[n]);
    {
        std::vector<int64_t> cur(fill.begin(), fill.end() - 1);
        for (int64_t j = c0; j < c1; ++j) {
            if (src[j] < 0 || dst[j] < 0) continue;
            raw[(size_t)cur[(size_t)src[j]]++] = (int)dst[j];
            raw[(size_t)cur[(size_t)dst[j]]++] = (int)src[j];
        }
    }
    // rows ascending without repeats, compacted in place: a row's distinct neighbours are found with one stamp per vertex and
    // emitted by a second counting pass over the TRANSPOSED stream (vertex ids ascending), so no per-row sort is needed
    std::vector<int> stamp(n, -1);
    std::vector<int64_t> cnt(n + 1, 0);
    for (size_t v = 0; v < n; ++v)
        for (int64_t p = fill[v]; p < fill[v + 1]; ++p) {
            const int w = raw[(size_t)p];
            if (stamp[(size_t)w] != (int)v) { stamp[(size_t)w] = (int)v; ++cnt[v + 1]; }
        }
    for (size_t v = 0; v < n; ++v) cnt[v + 1] += cnt[v];
    g.off.assign(cnt.begin(), cnt.end());
    g.nbr.resize((size_t)cnt[n]);
    {
        // the distinct pairs (v, w) are symmetric, so visiting w = 0 .. n-1 and appending w to the row of each of its distinct
        // neighbours v fills every row in ascending order
        std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
        std::fill(stamp.begin(), stamp.end(), -1);
        for (size_t w = 0; w < n; ++w)
            for (int64_t p = fill[w]; p < fill[w + 1]; ++p) {
                const int v = raw[(size_t)p];
                if (stamp[(size_t)v] != (int)w) { stamp[(size_t)v] = (int)w; g.nbr[(size_t)cur[(size_t)v]++] = (int)w; }
            }
    }
    if (g.n > 0 && g.n <= 4096) {
        g.words = (g.n + 63) >> 6;
        g.bits.assign(n * (size_t)g.words, 0ull);
        for (size_t v = 0; v < n; ++v)
            for (int64_t p = g.off[v]; p < g.off[v + 1]; ++p) {
                const int w = g.nbr[(size_t)p];
                g.bits[v * (size_t)g.words + ((size_t)w >> 6)] |= 1ull << (w & 63);
            }
    }
    return g;
}

// ---------------------------------------------------------------------------------------------------------------------
// Budget of one cut estimate (:184-205): h draws per member, a member's estimate counts only from `floor_hits` hits on.
// ---------------------------------------------------------------------------------------------------------------------
struct CutBudget { int draws; double floor_hits; };

inline CutBudget cut_budget(int k, double alpha, double beta, double delta) {
    const double want = 1.0 / (k * delta * alpha * alpha);
    const double d = want * want * std::log(k / beta);
    CutBudget b;
    b.draws = (std::isinf(d) || d > 100) ? 100 : (d < 10.0 ? 10 : static_cast<int>(std::ceil(d)));
    const double half = static_cast<double>(b.draws) * 0.5;
    b.floor_hits = want < half ? want : half;
    return b;
}

inline double fifth_power(int x) { return std::pow(static_cast<double>(x), 5.0); }      // the reference's bucket weight (:108, :159)

// ---------------------------------------------------------------------------------------------------------------------
// Ranking (the reference's APX-DD order, :52-168).  `pos[v]` = rank of v, `est[v]` = its bucket weight.
//   pass 1 (:82-125): ranks are visited in turn; the vertex at the rank draws h neighbours and counts those ranked behind it;
//       too few, and its score drops to 3*eta*deg and it is re-ranked among the vertices behind the cursor.  The order is
//       total (score descending, then id descending), so re-ranking is moving that one vertex to its place: everything between
//       moves up one rank.  The cursor then advances -- the vertex that moved INTO the cursor's rank is not visited (as in the
//       reference, whose loop index simply goes on).
//   pass 2 (:128-165): a vertex of degree <= k/eta keeps a weight only if k vertices can be reached from it through vertices
//       ranked behind it; the weight is then (neighbours ranked behind it)^5.
// `Draws` supplies below(n): the next integer in [0, n) of the backend's generator.
// ---------------------------------------------------------------------------------------------------------------------
struct Ranking { std::vector<int> seq, pos; std::vector<double> est; };

template <class Draws>
Ranking rank_vertices(const Csr &g, int k, double beta, Draws &rs) {
    const int n = g.n;
    const double eta = std::pow(beta, 1.0 / static_cast<double>(k - 1)) / (6.0 * k * k);
    const int h = static_cast<int>(std::ceil(10.0 / (eta * eta) * std::log(n)));
    Ranking R;
    R.seq.resize((size_t)n); R.pos.resize((size_t)n); R.est.assign((size_t)n, 0.0);
    std::vector<double> score((size_t)n);
    {   // initial ranks: degree descending, id descending inside a degree -- a counting sort
        int dmax = 0;
        for (int v = 0; v < n; ++v) { const int d = g.deg(v); score[(size_t)v] = static_cast<double>(d); if (d > dmax) dmax = d; }
        std::vector<int> first((size_t)dmax + 2, 0);
        for (int v = 0; v < n; ++v) ++first[(size_t)(dmax - g.deg(v)) + 1];
        for (int d = 0; d <= dmax; ++d) first[(size_t)d + 1] += first[(size_t)d];
        for (int v = n - 1; v >= 0; --v) { const int r = first[(size_t)(dmax - g.deg(v))]++; R.seq[(size_t)r] = v; R.pos[(size_t)v] = r; }
    }
    const double need = 2.0 * eta * h;
    for (int cursor = 0; cursor < n; ++cursor) {
        const int v = R.seq[(size_t)cursor];
        const int d = g.deg(v);
        if (d == 0) continue;                                           // weight stays 0, no draws
        const int *row = g.row(v);
        int behind = 0;
        for (int t = 0; t < h; ++t) behind += R.pos[(size_t)row[rs.below(d)]] > cursor ? 1 : 0;
        if (behind >= need) { R.est[(size_t)v] = fifth_power(d); continue; }
        R.est[(size_t)v] = 0.0;
        const double s = 3.0 * eta * d;
        score[(size_t)v] = s;
        int r = cursor;                                                 // slide v down past every vertex that now outranks it
        while (r + 1 < n) {
            const int x = R.seq[(size_t)r + 1];
            const double sx = score[(size_t)x];
            if (!(sx > s || (sx == s && x > v))) break;
            R.seq[(size_t)r] = x; R.pos[(size_t)x] = r;
            ++r;
        }
        R.seq[(size_t)r] = v; R.pos[(size_t)v] = r;
    }
    const double small = static_cast<double>(k) / eta;
    std::vector<int> mark((size_t)n, -1), ring((size_t)(k > 0 ? k : 1));
    for (int v = 0; v < n; ++v) {
        if (g.deg(v) > small) continue;
        const int pv = R.pos[(size_t)v];
        int head = 0, tail = 0, reached = 1;
        ring[(size_t)tail++] = v; mark[(size_t)v] = v;
        bool full = false;
        while (head < tail && reached < k && !full) {
            const int u = ring[(size_t)head++];
            const int *row = g.row(u);
            for (int t = 0, du = g.deg(u); t < du; ++t) {
                const int w = row[t];
                if (mark[(size_t)w] == v || R.pos[(size_t)w] <= pv) continue;
                mark[(size_t)w] = v;
                if (++reached >= k) { full = true; break; }
                ring[(size_t)tail++] = w;
            }
        }
        if (!full) { R.est[(size_t)v] = 0.0; continue; }
        int behind = 0;
        const int *row = g.row(v);
        for (int t = 0, dv = g.deg(v); t < dv; ++t) behind += R.pos[(size_t)row[t]] > pv ? 1 : 0;
        R.est[(size_t)v] = fifth_power(behind);
    }
    return R;
}

}  // namespace ugs_apx
