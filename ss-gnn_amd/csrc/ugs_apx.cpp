// ugs_apx.cpp -- `apx_ugs_sampler.sample_batch`, sequential backend.
//
// Contract (behaviour only): the reference's apx_ugs_sampler, src/samplers/apx_ugs_sampler/src/apx_ugs_sampler.cpp -- ranking
// (:52-168, ugs_apx_common.h), cut estimates (:175-236), growth (:243-312), growth probability over the orders of a vertex set
// (:318-382), rejection loop (:388-455), wrapper (:461-519).  The reference feeds EVERYTHING -- the ranking, every cut estimate of
// every trial of every sample -- from ONE sequential std::mt19937_64, so a bit-exact result has to consume that stream in the same
// sequence; that sequence and the floating-point expressions deciding comparisons are the contract, and the golden fixture
// tests/golden/apx_ugs.json (written by the unmodified reference build; generator committed beside the checker) holds this file to it.
//
// Everything else is this repo's own form: one `Trials` object owns flat scratch for the whole call (no container is allocated
// inside a trial); set membership is an epoch stamp per vertex; a cut estimate is a running sum handed back with its terms; the
// root and the growing member are found by bisection in prefix sums; a neighbour is drawn by counting the eligible ones first and
// walking to the drawn one; the orders of a vertex set come from a table of index permutations built once per k (factorial-base
// decoding, lexicographic, capped at 720 rows like the reference); libstdc++'s two distributions are restated explicitly
// (`Mt64`), so the result does not depend on the standard library the product is built against.
// Not part of the GPU hot path (ugs_sampler); the data-parallel variant of this sampler is ugs_apx_gpu.hip.
#include "../../include/ugs_mi355.h"

#include "ugs_apx_common.h"

#include <random>

namespace {
using namespace ugs_apx;

constexpr int kMembersMax = 64;              // k of this backend (the order table covers the last six members whatever k is)
constexpr int kOrdersMax = 720;              // reference :343
constexpr int kTrialsMax = 1000000;          // reference :411

// The reference's generator object (include/apx_ugs_sampler.hpp:12-24) is std::mt19937_64 behind libstdc++'s
// uniform_int_distribution<int>(0, n-1) and uniform_real_distribution<double>(0, 1).  Restated for a 64-bit engine:
//   below(n): Lemire's multiply-shift with rejection of the short tail (bits/uniform_int_dist.h, _S_nd): one engine word per
//             attempt, (word * n) >> 64 unless the low half falls under 2^64 mod n;
//   unit():   generate_canonical<double, 53> with one engine word: the word rounded to double, divided by 2^64, and the one value
//             that rounds up to 1.0 replaced by the largest double below it.
struct Mt64 {
    std::mt19937_64 engine;
    explicit Mt64(uint64_t seed) : engine(seed) {}
    int below(int n) {
        const uint64_t span = (uint64_t)(uint32_t)n;
        unsigned __int128 wide = (unsigned __int128)engine() * span;
        if ((uint64_t)wide < span) {
            const uint64_t tail = (0ull - span) % span;
            while ((uint64_t)wide < tail) wide = (unsigned __int128)engine() * span;
        }
        return (int)(uint64_t)(wide >> 64);
    }
    double unit() {
        const double x = static_cast<double>(engine()) / 18446744073709551616.0;
        return x < 1.0 ? x : 0x1.fffffffffffffp-1;
    }
};

// index permutations of `items` things in lexicographic order, at most kOrdersMax rows (row-major, one byte per entry).  Only the
// last min(items, 6) places vary within the first 720 rows (6! = 720); a row is decoded from its number in the factorial base.
struct OrderTable {
    int items = 0, rows = 0;
    std::vector<uint8_t> at;
    explicit OrderTable(int n) : items(n) {
        const int vary = n < 6 ? n : 6;
        rows = 1;
        for (int i = 2; i <= vary; ++i) rows *= i;
        if (rows > kOrdersMax) rows = kOrdersMax;
        at.resize((size_t)rows * (size_t)(n > 0 ? n : 1));
        for (int r = 0; r < rows; ++r) {
            uint8_t *out = at.data() + (size_t)r * (size_t)n;
            for (int i = 0; i < n - vary; ++i) out[i] = (uint8_t)i;
            uint8_t pool[6];
            for (int i = 0; i < vary; ++i) pool[i] = (uint8_t)(n - vary + i);
            int code = r, radix = rows;
            for (int left = vary; left > 0; --left) {
                radix /= left;                                        // (left - 1)!
                const int pickd = radix ? code / radix : 0;
                if (radix) code -= pickd * radix;
                out[n - left] = pool[pickd];
                for (int i = pickd; i + 1 < left; ++i) pool[i] = pool[i + 1];
            }
        }
    }
    const uint8_t *row(int r) const { return at.data() + (size_t)r * (size_t)items; }
};

class Trials {
  public:
    Trials(const Csr &g, const Ranking &o, int k, double epsilon, Mt64 &rs)
        : g_(g), o_(o), k_(k), rs_(rs), orders_(k - 1), stamp_((size_t)g.n, 0u), weight_to_((size_t)g.n) {
        beta_ = epsilon / 2.0;
        const double alpha = std::pow(beta_, 1.0 / static_cast<double>(k - 1)) / (6.0 * k * k * k);
        const double gamma = epsilon * std::pow(3.0, -k) * std::pow(k, static_cast<double>(-2));      // C2 = 2 (:397, :431, :440)
        grow_ = cut_budget(k, alpha, beta_, gamma / std::pow(k, 4.0));                                  // :253
        prob_ = cut_budget(k, alpha, beta_ / std::pow(k, 6.0), gamma / (k * k));                        // :358-359 (rho == gamma)
        double z = 0.0;
        for (int v = 0; v < g.n; ++v) { z += o.est[(size_t)v]; weight_to_[(size_t)v] = z; }          // the running sum the root draw compares with (:404-425)
        z_ = z;
        accept_scale_ = (beta_ / z_) * std::pow(k, static_cast<double>(-2));                          // C1 = 2 (:447)
    }

    bool any_weight() const { return z_ > 0.0; }

    // one sample: false when no trial below the cap is accepted (the reference then emits nothing for this sample)
    bool sample(int *out) {
        for (int trial = 0; trial < kTrialsMax; ++trial) {
            const int root = first_reaching(weight_to_.data(), g_.n, rs_.unit() * z_);
            if (o_.est[(size_t)root] <= 0.0) continue;
            if (!grow_from(root)) continue;
            const double p = growth_probability();
            if (p <= 0.0) continue;
            double a = accept_scale_ / (o_.est[(size_t)root] * p);
            if (a > 1.0) a = 1.0;
            if (rs_.unit() < a) {
                for (int j = 0; j < k_; ++j) out[j] = member_[j];
                return true;
            }
        }
        return false;
    }

  private:
    // first index whose running sum reaches x (the reference scans for `x <= sum`, defaulting to index 0); sums are non-decreasing
    static int first_reaching(const double *sums, int n, double x) {
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sums[mid] < x) lo = mid + 1; else hi = mid; }
        return lo < n ? lo : 0;
    }

    uint32_t fresh_epoch() {
        if (++epoch_ == 0u) { std::fill(stamp_.begin(), stamp_.end(), 0u); epoch_ = 1u; }
        return epoch_;
    }

    // cut estimate of the set set[0..count): per member, `draws` neighbours drawn with repetition; a draw hits when the neighbour is
    // ranked behind the root and outside the set; term = deg * hits / draws from `floor_hits` hits on.  Fills run_[] with the
    // running sums of the terms and returns the total.  Members without neighbours draw nothing.
    double estimate_cut(const int *set, int count, const CutBudget &b) {
        const uint32_t e = fresh_epoch();
        for (int i = 0; i < count; ++i) stamp_[(size_t)set[i]] = e;
        const int root_rank = o_.pos[(size_t)member_[0]];
        double total = 0.0;
        for (int i = 0; i < count; ++i) {
            const int u = set[i], d = g_.deg(u);
            double term = 0.0;
            if (d > 0) {
                const int *row = g_.row(u);
                int hits = 0;
                for (int t = 0; t < b.draws; ++t) {
                    const int w = row[rs_.below(d)];
                    hits += (o_.pos[(size_t)w] > root_rank && stamp_[(size_t)w] != e) ? 1 : 0;
                }
                if (hits >= b.floor_hits) term = static_cast<double>(d * hits) / static_cast<double>(b.draws);
            }
            total += term;
            run_[i] = total;
        }
        return total;
    }

    // growth from the root: k-1 times { estimate the cut of the members, pick a member in proportion to its term, pick one of its
    // neighbours ranked behind the root and not yet a member, uniformly }.  The last estimate's stamps still mark the members.
    bool grow_from(int root) {
        member_[0] = root;
        const int root_rank = o_.pos[(size_t)root];
        for (int size = 1; size < k_; ++size) {
            const double total = estimate_cut(member_, size, grow_);
            if (total <= 0.0) return false;
            const int from = member_[first_reaching(run_, size, rs_.unit() * total)];
            const uint32_t e = epoch_;
            const int *row = g_.row(from);
            const int d = g_.deg(from);
            int eligible = 0;
            for (int t = 0; t < d; ++t) eligible += (o_.pos[(size_t)row[t]] > root_rank && stamp_[(size_t)row[t]] != e) ? 1 : 0;
            if (eligible == 0) return false;
            int skip = rs_.below(eligible), next = -1;
            for (int t = 0; t < d; ++t) {
                const int w = row[t];
                if (o_.pos[(size_t)w] > root_rank && stamp_[(size_t)w] != e && skip-- == 0) { next = w; break; }
            }
            member_[size] = next;
        }
        return true;
    }

    // sum over the orders (root first, the other members in every lexicographic arrangement, at most 720) of the product over the
    // steps of (links from the newcomer into the prefix) / (estimated cut of the prefix); an order whose prefix has no cut adds 0
    // and ends there -- so do its draws
    double growth_probability() {
        int rest[kMembersMax];
        for (int j = 1; j < k_; ++j) {                               // the other members ascending (insertion: k is small)
            int x = member_[j], i = j - 1;
            while (i > 0 && rest[i - 1] > x) { rest[i] = rest[i - 1]; --i; }
            rest[i] = x;
        }
        int seq[kMembersMax];
        seq[0] = member_[0];
        double sum = 0.0;
        for (int r = 0; r < orders_.rows; ++r) {
            const uint8_t *ix = orders_.row(r);
            for (int j = 1; j < k_; ++j) seq[j] = rest[ix[j - 1]];
            double p = 1.0;
            for (int len = 1; len < k_; ++len) {
                const int newcomer = seq[len];
                int links = 0;
                for (int i = 0; i < len; ++i) links += g_.linked(seq[i], newcomer) ? 1 : 0;
                const double cut = estimate_cut(seq, len, prob_);
                if (!(cut > 0.0)) { p = 0.0; break; }
                p *= static_cast<double>(links) / cut;
            }
            sum += p;
        }
        return sum;
    }

    const Csr &g_;
    const Ranking &o_;
    const int k_;
    Mt64 &rs_;
    OrderTable orders_;
    std::vector<uint32_t> stamp_;
    std::vector<double> weight_to_;
    uint32_t epoch_ = 0u;
    double beta_ = 0.0, z_ = 0.0, accept_scale_ = 0.0;
    CutBudget grow_{}, prob_{};
    int member_[kMembersMax];
    double run_[kMembersMax];
};

}  // namespace

int ugs_internal_fail(int code, const char *msg);      // ugs_host.cpp: sets the message ugs_last_error() returns

extern "C" int ugs_apx_sample_batch(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr, int64_t ptr_len,
                                    int m_per_graph, int k, uint64_t seed, double epsilon, int64_t *samples_out, int64_t *num_samples_out) {
    if (!ptr || ptr_len < 2 || !num_samples_out || (num_cols > 0 && !edge_index)) return UGS_E_BAD_ARG;
    *num_samples_out = 0;
    if (k > kMembersMax) return ugs_internal_fail(UGS_E_UNSUPPORTED, "apx_ugs host backend: k <= 64");
    const int64_t c0 = ptr[0] > 0 ? ptr[0] : 0, c1 = ptr[1] < num_cols ? ptr[1] : num_cols;
    const Csr g = csr_of_columns(edge_index, edge_index + row_stride, c0, c1);
    if (g.n < k) return UGS_OK;
    Mt64 rs(seed);
    const Ranking o = rank_vertices(g, k, epsilon / 2.0, rs);
    Trials trials(g, o, k, epsilon, rs);
    if (!trials.any_weight()) return UGS_OK;
    int64_t got = 0;
    int row[kMembersMax];
    for (int s = 0; s < m_per_graph; ++s) {
        if (!trials.sample(row)) continue;
        if (samples_out) for (int j = 0; j < k; ++j) samples_out[got * k + j] = (int64_t)row[j];
        ++got;
    }
    *num_samples_out = got;
    return UGS_OK;
}
