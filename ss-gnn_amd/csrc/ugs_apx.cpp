// ugs_apx.cpp -- the `apx_ugs_sampler.sample_batch` entry point, host side.
//
// Contract: the reference's apx_ugs_sampler (src/samplers/apx_ugs_sampler/src/apx_ugs_sampler.cpp): APX-DD order (:52-168),
// EstimateCuts (:175-236), APX-RAND-GROW (:243-312), APX-PROB (:318-382), APX-UGS rejection loop (:388-455), wrapper (:461-519).
// The reference draws EVERYTHING -- the order, every cut estimate of every trial of every sample -- from ONE sequential
// std::mt19937_64 stream (:481-488), so its output has no parallel bit-exact form; this entry point is therefore kept as a
// host computation that consumes the same generator in the same sequence (std::mt19937_64 + libstdc++'s
// uniform_int_distribution / uniform_real_distribution, exactly as the reference's ApxRNG, include/apx_ugs_sampler.hpp:12-24)
// and is bit-exact with the reference on the same toolchain.  It is NOT part of the GPU hot path (ugs_sampler) and shares
// nothing with it.  Only the first graph is sampled and ptr[0]:ptr[1] is a range of edge COLUMNS (:15-33), as in the reference.
#include "../../include/ugs_mi355.h"

#include "ugs_apx_common.h"

#include <random>

namespace {
using namespace ugs_apx;

struct Stream {                       // one generator for everything, like the reference
    std::mt19937_64 gen;
    explicit Stream(uint64_t seed) : gen(seed) {}
    int below(int n) { return std::uniform_int_distribution<int>(0, n - 1)(gen); }
    double unit() { return std::uniform_real_distribution<double>(0.0, 1.0)(gen); }
};

// EstimateCuts: per vertex of U, the sampled number of neighbours after v in the order and outside U, scaled to its degree
std::vector<double> cut_estimates(const SimpleGraph &g, const Order &o, int v, const std::vector<int> &U, int k, double alpha,
                                  double beta, double delta, Stream &rs) {
    const double ell_raw = 1.0 / (k * delta * alpha * alpha);
    const double hd = ell_raw * ell_raw * std::log(k / beta);
    int h;
    if (std::isinf(hd) || hd > 100) h = 100;
    else if (hd < 10.0) h = 10;
    else h = static_cast<int>(std::ceil(hd));
    const double ell = std::min(ell_raw, static_cast<double>(h) * 0.5);
    std::vector<double> cuts(U.size(), 0.0);
    std::unordered_set<int> inU(U.begin(), U.end());
    for (size_t i = 0; i < U.size(); ++i) {
        const int u = U[i], d = g.deg(u);
        if (d == 0) { cuts[i] = 0.0; continue; }
        int hits = 0;
        for (int j = 0; j < h; ++j) {
            const int w = g.row(u)[rs.below(d)];
            if (o.pos[(size_t)v] < o.pos[(size_t)w] && inU.find(w) == inU.end()) ++hits;
        }
        cuts[i] = hits >= ell ? static_cast<double>(d * hits) / static_cast<double>(h) : 0.0;
    }
    return cuts;
}

std::vector<int> grow(const SimpleGraph &g, const Order &o, int v, int k, double alpha, double beta, double gamma, Stream &rs) {
    std::vector<int> S{v};
    const double delta = gamma / std::pow(k, 4.0);
    for (int i = 1; i < k; ++i) {
        const std::vector<double> cuts = cut_estimates(g, o, v, S, k, alpha, beta, delta, rs);
        double total = 0.0;
        for (double c : cuts) total += c;
        if (total <= 0.0) return {};
        const double r = rs.unit() * total;
        double run = 0.0;
        int from = S[0];
        for (size_t j = 0; j < S.size(); ++j) { run += cuts[j]; if (r <= run) { from = S[j]; break; } }
        std::vector<int> ok;
        for (int t = 0; t < g.deg(from); ++t) {
            const int w = g.row(from)[t];
            if (o.pos[(size_t)v] < o.pos[(size_t)w] && std::find(S.begin(), S.end(), w) == S.end()) ok.push_back(w);
        }
        if (ok.empty()) return {};
        S.push_back(ok[(size_t)rs.below((int)ok.size())]);
    }
    return S;
}

double growth_probability(const SimpleGraph &g, const Order &o, const std::vector<int> &S, double alpha, double beta, double rho, Stream &rs) {
    const int k = (int)S.size();
    if (k == 0) return 0.0;
    const int v = S[0];
    double total = 0.0;
    std::vector<int> rest(S.begin() + 1, S.end());
    std::sort(rest.begin(), rest.end());
    int perms = 0;
    do {
        std::vector<int> perm{v};
        perm.insert(perm.end(), rest.begin(), rest.end());
        double p = 1.0;
        for (int i = 0; i < k - 1; ++i) {
            std::vector<int> Si(perm.begin(), perm.begin() + i + 1);
            int links = 0;
            for (int u : Si) if (g.adjacent(u, perm[(size_t)i + 1])) ++links;
            const double delta = rho / (k * k);
            const std::vector<double> cuts = cut_estimates(g, o, v, Si, k, alpha, beta / std::pow(k, 6.0), delta, rs);
            double ci = 0.0;
            for (double c : cuts) ci += c;
            if (ci > 0.0) p *= static_cast<double>(links) / ci;
            else { p = 0.0; break; }
        }
        total += p;
        if (++perms >= 720) break;
    } while (std::next_permutation(rest.begin(), rest.end()));
    return total;
}

std::vector<int> one_sample(const SimpleGraph &g, const Order &o, int k, double epsilon, Stream &rs) {
    const int C1 = 2, C2 = 2;
    const double beta = epsilon / 2.0;
    const double alpha = std::pow(beta, 1.0 / static_cast<double>(k - 1)) / (6.0 * k * k * k);
    double Z = 0.0;
    for (int v = 0; v < g.n; ++v) Z += o.est[(size_t)v];
    if (Z <= 0.0) return {};
    for (int trial = 0; trial < 1000000; ++trial) {
        const double r = rs.unit() * Z;
        double run = 0.0;
        int v = 0;
        for (int u = 0; u < g.n; ++u) { run += o.est[(size_t)u]; if (r <= run) { v = u; break; } }
        if (o.est[(size_t)v] <= 0.0) continue;
        const double gamma = epsilon * std::pow(3.0, -k) * std::pow(k, static_cast<double>(-C2));
        const std::vector<int> S = grow(g, o, v, k, alpha, beta, gamma, rs);
        if (S.empty() || (int)S.size() != k) continue;
        const double rho = epsilon * std::pow(3.0, -k) * std::pow(k, static_cast<double>(-C2));
        const double p_hat = growth_probability(g, o, S, alpha, beta, rho, rs);
        if (p_hat <= 0.0) continue;
        double accept = (beta / Z) * std::pow(k, static_cast<double>(-C1)) / (o.est[(size_t)v] * p_hat);
        accept = std::min(1.0, accept);
        if (rs.unit() < accept) return S;
    }
    return {};
}

}  // namespace

extern "C" int ugs_apx_sample_batch(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr, int64_t ptr_len,
                                    int m_per_graph, int k, uint64_t seed, double epsilon, int64_t *samples_out, int64_t *num_samples_out) {
    if (!ptr || ptr_len < 2 || !num_samples_out || (num_cols > 0 && !edge_index)) return UGS_E_BAD_ARG;
    *num_samples_out = 0;
    const int64_t c0 = std::max<int64_t>(ptr[0], 0), c1 = std::min<int64_t>(ptr[1], num_cols);
    const SimpleGraph g = read_graph(edge_index, edge_index + row_stride, c0, c1);
    if (g.n < k) return UGS_OK;
    Stream rs(seed);
    const Order o = dominating_order(g, k, epsilon / 2.0, rs);
    int64_t got = 0;
    for (int s = 0; s < m_per_graph; ++s) {
        const std::vector<int> S = one_sample(g, o, k, epsilon, rs);
        if (S.empty()) continue;
        if (samples_out) for (int j = 0; j < k; ++j) samples_out[got * k + j] = (int64_t)S[(size_t)j];
        ++got;
    }
    *num_samples_out = got;
    return UGS_OK;
}
