// ugs_apx_gpu.hip -- GPU variant of `apx_ugs_sampler.sample_batch`.
//
// Contract: the reference's APX-UGS (src/samplers/apx_ugs_sampler/src/apx_ugs_sampler.cpp): per sample a rejection loop of up to
// 10^6 trials (:388-455), each trial = root draw proportional to the bucket estimates (:409-419), APX-RAND-GROW (:243-312) with
// EstimateCuts (:175-236), APX-PROB over the permutations of the grown set (:318-382), acceptance
// min(1, (beta/Z) k^-C1 / (est(v) p_hat)) (:440-447).  The reference draws every number of every trial of every sample from ONE
// sequential std::mt19937_64 stream, which has no parallel form (ugs_apx.cpp is that sequential restatement, bit-exact with it).
// Here the SAME algorithm runs with one generator per (sample, trial): every lane runs whole trials, all samples and thousands
// of trials at a time, and a sample's result is its accepted trial with the SMALLEST index -- a deterministic function of
// (graph, seed), independent of how trials are spread over lanes.  Parity with the reference is statistical: the output law is
// the same (the tests enumerate it exactly for k = 3: tests/test_apx_law.py pins the enumeration against the sequential
// restatement, tests/test_gpu_apx.py checks the GPU rows against it).  The APX-DD order (:52-168) is computed on the host (ugs_apx_common.h)
// with a counter-based generator.  Only the first graph is sampled and ptr[0]:ptr[1] is a range of edge COLUMNS, failed samples
// are dropped, as in the reference.
#include "../../include/ugs_mi355.h"
#include "ugs_apx_common.h"

#include <hip/hip_runtime.h>

#include <algorithm>

namespace {

constexpr int kMaxK = 32;                       // the product's k limit (UGS_KMAX); (k-1)! permutations per trial are capped at 720 like the reference (:370)
constexpr uint32_t kTrialCap = 1000000u;        // reference :411

struct ApxParams {
    const int64_t *off;       // [n+1]
    const int *nbr;           // sorted, duplicate-free rows
    const int *pos;           // position of every vertex in the APX-DD order
    const double *est;        // bucket estimate per vertex
    const double *cum;        // inclusive prefix sums of est in vertex order (root draw :413-418)
    int n, k;
    double Z, accept_scale;   // accept = min(1, accept_scale / (est(v) * p_hat)),  accept_scale = (beta / Z) * k^-C1
    int h_grow, h_prob;       // EstimateCuts sample counts of the two callers (:184-196)
    double ell_grow, ell_prob;
    uint64_t seed;
    uint32_t trial_cap;       // kTrialCap; UGS_APX_TRIAL_CAP lowers it (testing aid: large k makes a complete trial expensive)
};

struct TrialRng {             // one stream per (sample, trial): splitmix64 of the key, then xorshift64*
    uint64_t s;
    __device__ void init(uint64_t seed, uint32_t sample, uint32_t trial) {
        uint64_t z = seed + 0x9e3779b97f4a7c15ull * ((uint64_t)sample * 0x100000001b3ull + trial + 1ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31;
        s = z ? z : 0x2545f4914f6cdd1dull;
    }
    __device__ uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 2685821657736338717ull; }
    __device__ int below(int n) { return (int)__umulhi((uint32_t)(next() >> 32), (uint32_t)n); }      // uniform on [0, n)
    __device__ double unit() { return (double)(next() >> 11) * 0x1p-53; }                               // uniform on [0, 1)
};

__device__ bool adjacent(const ApxParams &P, int u, int w) {
    int64_t lo = P.off[u], hi = P.off[u + 1];
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; const int x = P.nbr[mid]; if (x == w) return true; if (x < w) lo = mid + 1; else hi = mid; }
    return false;
}

// EstimateCuts for U = S[0..nu): cuts[i] = deg(u) * hits / h if hits >= ell else 0 (reference :198-233); returns the total
__device__ double cut_estimates(const ApxParams &P, int v, const int *S, int nu, int h, double ell, TrialRng &rs, double *cuts) {
    const int pv = P.pos[v];
    double total = 0.0;
    for (int i = 0; i < nu; ++i) {
        const int u = S[i];
        const int64_t r0 = P.off[u];
        const int d = (int)(P.off[u + 1] - r0);
        double c = 0.0;
        if (d > 0) {
            int hits = 0;
            for (int j = 0; j < h; ++j) {
                const int w = P.nbr[r0 + rs.below(d)];
                bool out = pv < P.pos[w];
                for (int t = 0; t < nu; ++t) out = out && (S[t] != w);
                hits += out ? 1 : 0;
            }
            if ((double)hits >= ell) c = (double)(d * hits) / (double)h;
        }
        cuts[i] = c;
        total += c;
    }
    return total;
}

// one trial of the rejection loop; true = accepted, S[0..k) = the graphlet in growth order
__device__ bool run_trial(const ApxParams &P, uint32_t sample, uint32_t trial, int *S) {
    TrialRng rs;
    rs.init(P.seed, sample, trial);
    const int k = P.k;
    // root: first vertex whose running sum of estimates reaches r (:413-418)
    const double r = rs.unit() * P.Z;
    int lo = 0, hi = P.n - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (r <= P.cum[mid]) hi = mid; else lo = mid + 1; }
    const int v = lo;
    if (!(P.est[v] > 0.0)) return false;
    const int pv = P.pos[v];
    double cuts[kMaxK];
    // APX-RAND-GROW (:243-312)
    S[0] = v;
    for (int i = 1; i < k; ++i) {
        const double total = cut_estimates(P, v, S, i, P.h_grow, P.ell_grow, rs, cuts);
        if (!(total > 0.0)) return false;
        const double rr = rs.unit() * total;
        double run = 0.0;
        int from = S[0];
        for (int j = 0; j < i; ++j) { run += cuts[j]; if (rr <= run) { from = S[j]; break; } }
        // neighbours of `from` after v in the order and outside S, in row order
        int nok = 0;
        for (int64_t p = P.off[from]; p < P.off[from + 1]; ++p) {
            const int w = P.nbr[p];
            bool ok = pv < P.pos[w];
            for (int t = 0; t < i; ++t) ok = ok && (S[t] != w);
            nok += ok ? 1 : 0;
        }
        if (nok == 0) return false;
        int pick = rs.below(nok), chosen = -1;
        for (int64_t p = P.off[from]; p < P.off[from + 1] && chosen < 0; ++p) {
            const int w = P.nbr[p];
            bool ok = pv < P.pos[w];
            for (int t = 0; t < i; ++t) ok = ok && (S[t] != w);
            if (ok) { if (pick == 0) chosen = w; --pick; }
        }
        S[i] = chosen;
    }
    // APX-PROB (:318-382): permutations of the non-root vertices in lexicographic order from the sorted one, at most 720
    int perm[kMaxK];
    perm[0] = v;
    for (int i = 1; i < k; ++i) perm[i] = S[i];
    for (int i = 2; i < k; ++i) { const int x = perm[i]; int j = i; while (j > 1 && perm[j - 1] > x) { perm[j] = perm[j - 1]; --j; } perm[j] = x; }
    double p_hat = 0.0;
    for (int count = 0;;) {
        double p = 1.0;
        for (int i = 0; i < k - 1; ++i) {
            int links = 0;
            for (int t = 0; t <= i; ++t) links += adjacent(P, perm[t], perm[i + 1]) ? 1 : 0;
            const double ci = cut_estimates(P, v, perm, i + 1, P.h_prob, P.ell_prob, rs, cuts);
            if (ci > 0.0) p *= (double)links / ci;
            else { p = 0.0; break; }
        }
        p_hat += p;
        if (++count >= 720) break;
        // std::next_permutation on perm[1..k)
        int i = k - 2;
        while (i >= 1 && perm[i] >= perm[i + 1]) --i;
        if (i < 1) break;
        int j = k - 1;
        while (perm[j] <= perm[i]) --j;
        { const int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        for (int a = i + 1, b = k - 1; a < b; ++a, --b) { const int t = perm[a]; perm[a] = perm[b]; perm[b] = t; }
    }
    if (!(p_hat > 0.0)) return false;
    double accept = P.accept_scale / (P.est[v] * p_hat);
    accept = accept < 1.0 ? accept : 1.0;
    return rs.unit() < accept;
}

// Every lane runs whole trials of its sample (blockIdx.y) until a smaller accepted trial is known: best[s] ends as the smallest
// accepted trial index below the cap, whatever the scheduling (a lane only skips trials that can no longer win).
__global__ __launch_bounds__(256) void ugs_apx_trials(ApxParams P, int sample0, unsigned int *best) {
    const uint32_t s = (uint32_t)(sample0 + (int)blockIdx.y);
    const uint32_t stride = gridDim.x * 256u;
    int S[kMaxK];
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < P.trial_cap; t += stride) {
        if (t >= __atomic_load_n(&best[blockIdx.y], __ATOMIC_RELAXED)) break;
        if (run_trial(P, s, t, S)) atomicMin(&best[blockIdx.y], t);
    }
}

// the winning trial again, one lane per sample, to write its graphlet
__global__ __launch_bounds__(64) void ugs_apx_emit(ApxParams P, int sample0, int count, const unsigned int *best, int64_t *out /* [count, k] */) {
    const int i = (int)(blockIdx.x * 64 + threadIdx.x);
    if (i >= count) return;
    const unsigned int t = best[i];
    if (t >= P.trial_cap) return;
    int S[kMaxK];
    const bool ok = run_trial(P, (uint32_t)(sample0 + i), t, S);
    for (int j = 0; j < P.k; ++j) out[(int64_t)i * P.k + j] = ok ? (int64_t)S[j] : (int64_t)-1;
}

struct CounterRng {           // host generator of the APX-DD order: one draw per call, keyed by the seed and a running counter
    uint64_t seed, ctr = 0;
    explicit CounterRng(uint64_t s) : seed(s) {}
    int below(int n) {
        uint64_t z = seed + 0x9e3779b97f4a7c15ull * (++ctr);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31;
        return (int)(((z >> 32) * (uint64_t)n) >> 32);
    }
};

}  // namespace

int ugs_internal_fail(int code, const char *msg);      // ugs_host.cpp: sets the message ugs_last_error() returns
int ugs_internal_ctx(int *device, hipStream_t *stream);   // ugs_host.cpp: the calling thread's device and stream (ugs_set_device / ugs_set_stream)

extern "C" int ugs_apx_gpu_sample_batch(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr, int64_t ptr_len,
                                        int m_per_graph, int k, uint64_t seed, double epsilon, int64_t *samples_out, int64_t *num_samples_out,
                                        int32_t *order_pos_out, double *est_out, int64_t order_capacity) {
    using namespace ugs_apx;
    auto fail = [](int code, const char *msg) { return ugs_internal_fail(code, msg); };
    if (!ptr || ptr_len < 2 || !num_samples_out || (num_cols > 0 && !edge_index)) return fail(UGS_E_BAD_ARG, "bad arguments to apx sample_batch");
    if (k < 2 || k > kMaxK) return fail(UGS_E_UNSUPPORTED, "the GPU variant of apx_ugs supports 2 <= k <= 32");
    if (!(epsilon > 0.0)) return fail(UGS_E_BAD_ARG, "epsilon must be > 0");
    *num_samples_out = 0;
    int dev_id = 0;
    hipStream_t st = nullptr;
    if (int rc = ugs_internal_ctx(&dev_id, &st)) return rc;          // the thread's device (ugs_set_device) and stream, like every other entry point
    const int64_t c0 = std::max<int64_t>(ptr[0], 0), c1 = std::min<int64_t>(ptr[1], num_cols);
    const Csr g = csr_of_columns(edge_index, edge_index + row_stride, c0, c1);
    if (g.n < k || m_per_graph <= 0) return UGS_OK;
    CounterRng ors(seed ^ 0xa0761d6478bd642full);
    const double beta = epsilon / 2.0;
    const Ranking o = rank_vertices(g, k, beta, ors);
    if (order_pos_out && est_out && order_capacity >= g.n)
        for (int v = 0; v < g.n; ++v) { order_pos_out[v] = o.pos[(size_t)v]; est_out[v] = o.est[(size_t)v]; }
    std::vector<double> cum((size_t)g.n);
    double Z = 0.0;
    for (int v = 0; v < g.n; ++v) { Z += o.est[(size_t)v]; cum[(size_t)v] = Z; }
    if (!(Z > 0.0)) return UGS_OK;
    // constants of one_sample (:392-447)
    const int C1 = 2, C2 = 2;
    const double alpha = std::pow(beta, 1.0 / static_cast<double>(k - 1)) / (6.0 * k * k * k);
    const double gamma = epsilon * std::pow(3.0, -k) * std::pow(k, static_cast<double>(-C2));
    const double rho = gamma;
    ApxParams P{};
    const CutBudget bg = cut_budget(k, alpha, beta, gamma / std::pow(k, 4.0)), bp = cut_budget(k, alpha, beta / std::pow(k, 6.0), rho / (k * k));
    P.h_grow = bg.draws; P.ell_grow = bg.floor_hits; P.h_prob = bp.draws; P.ell_prob = bp.floor_hits;
    P.n = g.n; P.k = k; P.Z = Z; P.seed = seed;
    P.trial_cap = kTrialCap;
    if (const char *ev = std::getenv("UGS_APX_TRIAL_CAP")) { const long c = std::atol(ev); if (c > 0 && c < (long)kTrialCap) P.trial_cap = (uint32_t)c; }
    P.accept_scale = (beta / Z) * std::pow(k, static_cast<double>(-C1));
    // device copies
    std::vector<int> pos32(o.pos.begin(), o.pos.end());
    const size_t b_off = (size_t)(g.n + 1) * sizeof(int64_t), b_nbr = std::max<size_t>(g.nbr.size(), 1) * sizeof(int), b_pos = (size_t)g.n * sizeof(int),
                 b_d = (size_t)g.n * sizeof(double);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_nbr = up(b_off), o_pos = o_nbr + up(b_nbr), o_est = o_pos + up(b_pos), o_cum = o_est + up(b_d), o_best = o_cum + up(b_d),
                 o_out = o_best + up((size_t)m_per_graph * sizeof(unsigned int)), total = o_out + up((size_t)m_per_graph * k * sizeof(int64_t));
    char *d = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d), total);
    if (e != hipSuccess) return fail(UGS_E_HIP, hipGetErrorString(e));
    auto bail = [&](hipError_t err) { (void)hipFree(d); return ugs_internal_fail(UGS_E_HIP, hipGetErrorString(err)); };
    if ((e = hipMemcpy(d, g.off.data(), b_off, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
    if (!g.nbr.empty() && (e = hipMemcpy(d + o_nbr, g.nbr.data(), g.nbr.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
    if ((e = hipMemcpy(d + o_pos, pos32.data(), b_pos, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
    if ((e = hipMemcpy(d + o_est, o.est.data(), b_d, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
    if ((e = hipMemcpy(d + o_cum, cum.data(), b_d, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
    if ((e = hipMemsetAsync(d + o_best, 0xFF, (size_t)m_per_graph * sizeof(unsigned int), st)) != hipSuccess) return bail(e);
    if ((e = hipMemsetAsync(d + o_out, 0xFF, (size_t)m_per_graph * k * sizeof(int64_t), st)) != hipSuccess) return bail(e);
    P.off = reinterpret_cast<const int64_t *>(d); P.nbr = reinterpret_cast<const int *>(d + o_nbr); P.pos = reinterpret_cast<const int *>(d + o_pos);
    P.est = reinterpret_cast<const double *>(d + o_est); P.cum = reinterpret_cast<const double *>(d + o_cum);
    unsigned int *best = reinterpret_cast<unsigned int *>(d + o_best);
    int64_t *out = reinterpret_cast<int64_t *>(d + o_out);
    // all samples side by side: grid.y = samples of a slab, grid.x * 256 lanes run the trials of one sample
    const int slab = 4096;
    for (int s0 = 0; s0 < m_per_graph; s0 += slab) {
        const int ns = std::min(slab, m_per_graph - s0);
        int gx = std::max(1, std::min(64, 8192 / ns));
        hipLaunchKernelGGL(ugs_apx_trials, dim3((unsigned)gx, (unsigned)ns), dim3(256), 0, st, P, s0, best + s0);
        hipLaunchKernelGGL(ugs_apx_emit, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, st, P, s0, ns, best + s0, out + (int64_t)s0 * k);
        if ((e = hipGetLastError()) != hipSuccess) return bail(e);
    }
    std::vector<int64_t> h_out((size_t)m_per_graph * k);
    if ((e = hipMemcpyAsync(h_out.data(), out, h_out.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st)) != hipSuccess) return bail(e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return bail(e);
    (void)hipFree(d);
    int64_t got = 0;
    for (int s = 0; s < m_per_graph; ++s) {
        if (h_out[(size_t)s * k] < 0) continue;                // no accepted trial below the cap: the sample is dropped (:450-453)
        if (samples_out) for (int j = 0; j < k; ++j) samples_out[got * k + j] = h_out[(size_t)s * k + j];
        ++got;
    }
    *num_samples_out = got;
    return UGS_OK;
}
