// ugs_host.cpp -- host side of libugs_mi355.so: graph preprocessing, handle registry, the batch LRU, device plans and
// the C ABI of include/ugs_mi355.h.  Sampling itself runs only in the gfx950 kernels of ugs_kernels.hip; there is no
// CPU sampling path in this library.
//
// Behavioural contract (what must come out bit-identical): the reference `ugs_sampler`
// (AniruddhaMandal/SS-GNN src/samplers/ugs_sampler): preprocessing src/preproc.cpp:32-256 + include/sampler.hpp:39-69,
// handle registry src/preproc.cpp:262-314, LRU + graph hash include/cache.hpp:15-109, batch wrapper
// src/ugs_sampler_batch_extension.cpp:41-299.  The data structures here are flat arrays laid out for the GPU
// (see ugs_device.h), not the reference's.
#include "../../include/ugs_mi355.h"
#include "ugs_device.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <thread>
#include <vector>

struct ugs_plan;
namespace { void plan_unref(ugs_plan *p); void arena_release(int dev, int64_t roots_off, int64_t n_roots, int64_t via_off, int64_t n_via); void pin_slot_put(char *p); }

// launcher of the epsilon_uniform kernels (ugs_eps.hip)
struct UgsEpsLaunch {
    const UgsGraphDesc *graphs; const int64_t *rowptr; const int32_t *nbr; const int32_t *ecs; int64_t num_graphs;
    int32_t m, k, mode, max_attempts; uint64_t seed; double epsilon; int64_t rows;
    int64_t *nodes; uint32_t *counts; const int64_t *edge_ptr; int64_t *edge_index; int64_t *edge_src; int64_t ld;
};
hipError_t ugs_eps_launch(const UgsEpsLaunch &l, int fill, int cus, hipStream_t s);

// device-side preprocessing stages (ugs_preproc.hip)
struct UgsDevPre;
size_t ugs_devpre_bytes(int64_t n, int64_t E);
hipError_t ugs_devpre_csr(UgsDevPre **out, const int64_t *h_src, const int64_t *h_dst, int64_t E, int64_t n, hipStream_t s, int64_t *h_rowptr, int64_t *nnz_out);
hipError_t ugs_devpre_roots(UgsDevPre *d, const int32_t *h_order, const int32_t *h_rank, int k, int32_t *h_sdeg, uint8_t *h_reach);
hipError_t ugs_devpre_download(UgsDevPre *d, int32_t *h_nbr, int32_t *h_col);
void ugs_devpre_free(UgsDevPre *d);
void ugs_devpre_trim(UgsDevPre *d);
size_t ugs_devpre_resident_bytes(const UgsDevPre *d);
hipError_t ugs_devpre_assemble(UgsDevPre *d, const int64_t *h_colmap, int64_t cols, int2 *adj, int2 *adjf, hipStream_t s);

namespace {

thread_local std::string t_err;
int fail(int code, const std::string &msg) { t_err = msg; return code; }
}  // namespace
// the other translation units of the library report through the same thread-local message (internal, not part of the C ABI)
int ugs_internal_fail(int code, const char *msg) { return fail(code, msg ? msg : ""); }
int ugs_internal_ctx(int *device, hipStream_t *stream);    // below: the calling thread's device and stream (ugs_set_device / ugs_set_stream)
namespace {
int fail_hip(hipError_t e, const char *what) { return fail(UGS_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail_hip(e_, #expr); } while (0)

struct Lap {   // UGS_DEBUG=1: wall-clock of the host stages of a large preprocessing / plan build
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double operator()() { auto n = std::chrono::steady_clock::now(); double d = std::chrono::duration<double>(n - t).count(); t = n; return d; }
};

bool debug_on() { static const bool on = [] { const char *e = std::getenv("UGS_DEBUG"); return e && std::string(e) == "1"; }(); return on; }

// ---------------------------------------------------------------------------------------------------------------
// per-graph preprocessing result (host copy; the device plan is assembled from these)
// ---------------------------------------------------------------------------------------------------------------
struct Graph {
    int64_t n = 0, nnz = 0;
    int k_built = 0;
    std::vector<int64_t> rowptr;      // n+1, 0-based
    std::vector<int32_t> nbr, col;    // CSR neighbours and source column of each entry
    std::vector<int32_t> order, rank; // order[vi] = vertex, rank[vertex] = vi
    std::vector<int32_t> sdeg;        // suffix degree per order position
    std::vector<uint8_t> reach;       // scratch: root reaches k vertices inside its suffix graph
    std::vector<double> weight;       // bucket weight per order position
    std::vector<double> prob;         // alias table (only if Z > 0)
    std::vector<int32_t> alias;
    double Z = 0.0;
    int nonzero = 0;
    int level = 0;                    // relaxation level of the root draw
    std::vector<int32_t> viable;      // order positions for levels 1, 2
    int64_t n_viable = 0;             // their number (a graph preprocessed by the device batch pass has only this)
    // A graph the device batch pass preprocessed (cold path, ugs_bp_roots) exists on the host as a STUB: n, nnz, level, Z, the degree
    // statistics and its root records in a device's arena -- no arrays.  The general path, which has the graph's columns at hand
    // whenever it meets the graph, fills the arrays in then (complete_on_host).
    bool host_ready = true;
    std::vector<int64_t> stub_cols;   // stub: its span of the batch's columns, sources then targets (global ids), and
    int64_t stub_lo = 0;              //       the batch's node offset of the graph -- what completion needs, dropped afterwards
    int64_t max_deg = 0;
    double sb_deg = 0.0;              // size-biased mean CSR degree (sum d^2 / sum d)
    std::mutex plan_mu;
    ugs_plan *plan = nullptr;         // device-resident copy for the handle API, built on first sample()
    std::mutex pre_mu;
    UgsDevPre *devpre = nullptr;      // device CSR left by preprocess_on_device for the FIRST plan assembled from this graph
    int devpre_dev = -1;
    UgsDevPre *take_devpre(int dev) { std::lock_guard<std::mutex> lk(pre_mu); if (!devpre || devpre_dev != dev) return nullptr; UgsDevPre *d = devpre; devpre = nullptr; return d; }
    // root records / viable list of this graph resident in a device's arena (device batch pass: plans of new combinations of
    // known graphs point at them instead of copying them); offsets in elements, -1 = none
    struct DevRoots { int dev; int64_t roots_off, n_roots, via_off, n_via; };
    std::vector<DevRoots> dev_roots;
    ~Graph() {
        if (plan) plan_unref(plan);
        if (devpre) ugs_devpre_free(devpre);
        for (auto &d : dev_roots) arena_release(d.dev, d.roots_off, d.n_roots, d.via_off, d.n_via);
    }
};

// CSR of the symmetrised multigraph, entries in column order (both endpoints of column j, u's row first).
void build_adjacency(Graph &G, const int64_t *src, const int64_t *dst, int64_t E) {
    const int64_t n = G.n;
    G.rowptr.assign((size_t)n + 1, 0);
    for (int64_t j = 0; j < E; ++j) {
        const int64_t u = src[j], v = dst[j];
        if ((uint64_t)u >= (uint64_t)n || (uint64_t)v >= (uint64_t)n) continue;     // silently skipped columns
        ++G.rowptr[(size_t)u + 1];
        ++G.rowptr[(size_t)v + 1];
    }
    for (int64_t r = 0; r < n; ++r) G.rowptr[(size_t)r + 1] += G.rowptr[(size_t)r];
    G.nnz = G.rowptr[(size_t)n];
    G.nbr.resize((size_t)G.nnz);
    G.col.resize((size_t)G.nnz);
    std::vector<int64_t> wr(G.rowptr.begin(), G.rowptr.end() - 1);
    for (int64_t j = 0; j < E; ++j) {
        const int64_t u = src[j], v = dst[j];
        if ((uint64_t)u >= (uint64_t)n || (uint64_t)v >= (uint64_t)n) continue;
        int64_t a = wr[(size_t)u]++;
        G.nbr[(size_t)a] = (int32_t)v; G.col[(size_t)a] = (int32_t)j;
        int64_t b = wr[(size_t)v]++;
        G.nbr[(size_t)b] = (int32_t)u; G.col[(size_t)b] = (int32_t)j;
    }
}

// The reference's "remove the max-degree vertex, then reverse" ordering is the stable ascending sort by
// (CSR degree, vertex id): every vertex is popped from the bucket of its ORIGINAL degree (later, lower-degree
// duplicates are stale when reached), buckets drain high to low and back-to-front.  One counting sort.
void order_by_degree(Graph &G) {
    const int64_t n = G.n;
    G.order.resize((size_t)n);
    G.rank.resize((size_t)n);
    int64_t maxd = 0;
    double s1 = 0, s2 = 0;
    for (int64_t v = 0; v < n; ++v) {
        int64_t d = G.rowptr[(size_t)v + 1] - G.rowptr[(size_t)v];
        maxd = std::max(maxd, d);
        s1 += (double)d; s2 += (double)d * (double)d;
    }
    G.max_deg = maxd;
    G.sb_deg = s1 > 0 ? s2 / s1 : 0.0;
    std::vector<int64_t> start((size_t)maxd + 2, 0);
    for (int64_t v = 0; v < n; ++v) ++start[(size_t)(G.rowptr[(size_t)v + 1] - G.rowptr[(size_t)v]) + 1];
    for (int64_t d = 0; d <= maxd; ++d) start[(size_t)d + 1] += start[(size_t)d];
    for (int64_t v = 0; v < n; ++v) {
        int64_t d = G.rowptr[(size_t)v + 1] - G.rowptr[(size_t)v];
        int64_t pos = start[(size_t)d]++;
        G.order[(size_t)pos] = (int32_t)v;
        G.rank[(size_t)v] = (int32_t)pos;
    }
}

// Vose alias table, stacks filled in ascending index; order-sensitive IEEE arithmetic kept exactly
// (p = w*n/sum; p[l] = (p[l] + p[s]) - 1.0).
void build_alias(Graph &G) {
    const int n = (int)G.n;
    G.prob.assign((size_t)n, 0.0);
    G.alias.assign((size_t)n, 0);
    if (n == 0) return;
    double total = 0.0;
    for (int i = 0; i < n; ++i) total += G.weight[(size_t)i];
    const double denom = total > 0 ? total : 1.0;
    std::vector<double> p((size_t)n);
    std::vector<int32_t> lo, hi;
    lo.reserve((size_t)n); hi.reserve((size_t)n);
    for (int i = 0; i < n; ++i) {
        p[(size_t)i] = G.weight[(size_t)i] * n / denom;
        (p[(size_t)i] < 1.0 ? lo : hi).push_back(i);
    }
    while (!lo.empty() && !hi.empty()) {
        const int32_t s = lo.back(); lo.pop_back();
        const int32_t l = hi.back();
        G.prob[(size_t)s] = p[(size_t)s];
        G.alias[(size_t)s] = l;
        p[(size_t)l] = (p[(size_t)l] + p[(size_t)s]) - 1.0;
        if (p[(size_t)l] < 1.0) { hi.pop_back(); lo.push_back(l); }
    }
    for (int32_t i : hi) G.prob[(size_t)i] = 1.0;
    for (int32_t i : lo) G.prob[(size_t)i] = 1.0;
}

// suffix degrees and k-reachability of every root inside its suffix graph (host form; ugs_preproc.hip is the device form)
void root_stats_host(Graph &G, int k) {
    const int n = (int)G.n;
    G.sdeg.assign((size_t)n, 0);
    for (int vi = 0; vi < n; ++vi) {
        const int32_t v = G.order[(size_t)vi];
        int32_t c = 0;
        for (int64_t p = G.rowptr[(size_t)v]; p < G.rowptr[(size_t)v + 1]; ++p) c += G.rank[(size_t)G.nbr[(size_t)p]] >= vi;
        G.sdeg[(size_t)vi] = c;
    }
    G.reach.assign((size_t)n, 0);
    std::vector<int32_t> mark((size_t)n, -1), reached;
    reached.reserve((size_t)std::max(k, 1) + 1);
    for (int vi = 0; vi < n; ++vi) {
        reached.clear();
        reached.push_back(G.order[(size_t)vi]);
        mark[(size_t)reached[0]] = vi;
        for (size_t h = 0; h < reached.size() && (int)reached.size() < k; ++h) {
            const int32_t u = reached[h];
            for (int64_t p = G.rowptr[(size_t)u]; p < G.rowptr[(size_t)u + 1] && (int)reached.size() < k; ++p) {
                const int32_t w = G.nbr[(size_t)p];
                if (G.rank[(size_t)w] < vi || mark[(size_t)w] == vi) continue;
                mark[(size_t)w] = vi;
                reached.push_back(w);
            }
        }
        G.reach[(size_t)vi] = (int)reached.size() >= k;
    }
}

// bucket weights d^(k-1), Z (summed in order position order: the floating-point result depends on it), alias table
void weigh_roots(Graph &G, int k) {
    const int n = (int)G.n;
    G.weight.assign((size_t)n, 0.0);
    G.Z = 0.0;
    G.nonzero = 0;
    for (int vi = 0; vi < n; ++vi) {
        if (G.reach[(size_t)vi]) {
            const double d = (double)std::max<int32_t>(1, G.sdeg[(size_t)vi]);
            double b = 1.0;
            for (int t = 1; t < k; ++t) b *= d;
            G.weight[(size_t)vi] = b;
            G.Z += b;
            if (b > 0.0) ++G.nonzero;
        }
    }
    std::vector<uint8_t>().swap(G.reach);
    if (G.Z > 0.0) build_alias(G); else { G.prob.assign((size_t)n, 0.0); G.alias.assign((size_t)n, 0); }
    // relaxation levels of the root draw
    G.viable.clear();
    if (G.nonzero > 0) G.level = 0;
    else {
        G.level = 1;
        for (int vi = 0; vi < n; ++vi) if (G.sdeg[(size_t)vi] > 0) G.viable.push_back(vi);
        if (G.viable.empty()) { G.level = 2; for (int vi = 0; vi < n; ++vi) G.viable.push_back(vi); }
    }
    G.n_viable = (int64_t)G.viable.size();
    if (debug_on()) std::fprintf(stderr, "[UGS PREPROC] n=%d k=%d Z=%.2e viable=%d/%d\n", n, k, G.Z, G.nonzero, n);
}

// arrays of a stub (see Graph::host_ready) from the graph's columns, with the k it was preprocessed for: the same values the
// device computed (both equal the reference's), now on the host as well
void complete_on_host(Graph &G) {
    std::lock_guard<std::mutex> lk(G.pre_mu);
    if (G.host_ready) return;
    const size_t span = G.stub_cols.size() / 2;
    std::vector<int64_t> ru, rv;
    ru.reserve(span); rv.reserve(span);
    for (size_t j = 0; j < span; ++j) {                                  // the graph's columns: both endpoints inside its node range
        const int64_t u = G.stub_cols[j] - G.stub_lo, v = G.stub_cols[span + j] - G.stub_lo;
        if ((uint64_t)u < (uint64_t)G.n && (uint64_t)v < (uint64_t)G.n) { ru.push_back(u); rv.push_back(v); }
    }
    build_adjacency(G, ru.data(), rv.data(), (int64_t)ru.size());
    order_by_degree(G);
    root_stats_host(G, G.k_built);
    weigh_roots(G, G.k_built);
    std::vector<int64_t>().swap(G.stub_cols);
    G.host_ready = true;
}

int preprocess_on_device(Graph &G, const int64_t *src, const int64_t *dst, int64_t E, int k, bool &done);   // below, after the device helpers

int make_graph(const int64_t *src, const int64_t *dst, int64_t E, int64_t n, int k, std::shared_ptr<Graph> &out) {
    if (n < 0) return fail(UGS_E_BAD_ARG, "num_nodes must be >= 0");
    if (n >= ((int64_t)1 << 30) - 1 || E >= (int64_t)INT32_MAX) return fail(UGS_E_UNSUPPORTED, "graph too large: num_nodes must be < 2^30 - 1 and columns < 2^31 - 1");
    auto G = std::make_shared<Graph>();
    G->n = n;
    G->k_built = k;
    bool on_device = false;
    if (int rc = preprocess_on_device(*G, src, dst, E, k, on_device)) return rc;
    if (!on_device) {
        Lap lap;
        build_adjacency(*G, src, dst, E);
        if (G->nnz >= (int64_t)INT32_MAX) return fail(UGS_E_UNSUPPORTED, "graph too large: CSR entries must be < 2^31 - 1");
        const double t_csr = lap();
        order_by_degree(*G);
        const double t_order = lap();
        root_stats_host(*G, k);
        if (debug_on() && E >= ((int64_t)1 << 21))
            std::fprintf(stderr, "[UGS PREPROC] n=%lld columns=%lld on the host: CSR %.3fs, degree order %.3fs, suffix degrees+reachability %.3fs\n",
                         (long long)n, (long long)E, t_csr, t_order, lap());
    }
    Lap lap;
    weigh_roots(*G, k);
    if (debug_on() && E >= ((int64_t)1 << 21)) std::fprintf(stderr, "[UGS PREPROC] weights, Z, alias table (host) %.3fs\n", lap());
    out = std::move(G);
    return UGS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// handle registry (monotonically increasing int64 handles from 1)
// ---------------------------------------------------------------------------------------------------------------
// The process-wide containers below are heap objects that are never destroyed: at process exit no destructor of this library
// runs (a Graph's destructor releases device plans into the scratch pool and calls the HIP runtime, neither of which may
// still exist during static destruction).
std::mutex g_reg_mu;
std::unordered_map<int64_t, std::shared_ptr<Graph>> &g_reg = *new std::unordered_map<int64_t, std::shared_ptr<Graph>>();
int64_t g_next_handle = 1;

std::shared_ptr<Graph> lookup(int64_t h) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_reg.find(h);
    return it == g_reg.end() ? nullptr : it->second;
}
int64_t enroll(std::shared_ptr<Graph> g) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    int64_t h = g_next_handle++;
    g_reg[h] = std::move(g);
    return h;
}
void drop(int64_t h) { std::lock_guard<std::mutex> lk(g_reg_mu); g_reg.erase(h); }

// ---------------------------------------------------------------------------------------------------------------
// batch LRU: graph hash -> handle.  Capacity from UGS_CACHE_SIZE (default 1000); capacity 0 never evicts.
// The key deliberately matches the reference's (FNV-1a over n, #cols and (strided) renumbered columns, NOT k), so
// that the same call history produces the same reuse -- including reuse of weights built for another k.
// ---------------------------------------------------------------------------------------------------------------
struct Lru {
    size_t capacity = 1000;
    std::list<std::pair<uint64_t, int64_t>> items;     // front = most recent
    std::unordered_map<uint64_t, std::list<std::pair<uint64_t, int64_t>>::iterator> index;
    int64_t hits = 0, misses = 0;
    bool get(uint64_t key, int64_t &val) {
        auto it = index.find(key);
        if (it == index.end()) return false;
        items.splice(items.begin(), items, it->second);
        val = it->second->second;
        return true;
    }
    bool put(uint64_t key, int64_t val, int64_t &evicted) {   // true if something was evicted
        auto it = index.find(key);
        if (it != index.end()) { it->second->second = val; items.splice(items.begin(), items, it->second); return false; }
        bool ev = false;
        if (capacity > 0 && items.size() >= capacity) {
            evicted = items.back().second;
            index.erase(items.back().first);
            items.pop_back();
            ev = true;
        }
        items.emplace_front(key, val);
        index[key] = items.begin();
        return ev;
    }
    void erase(uint64_t key) {
        auto it = index.find(key);
        if (it == index.end()) return;
        items.erase(it->second);
        index.erase(it);
    }
};
std::mutex g_lru_mu;
Lru *g_lru = nullptr;
Lru &lru() {   // call with g_lru_mu held
    if (!g_lru) {
        g_lru = new Lru();
        if (const char *e = std::getenv("UGS_CACHE_SIZE")) g_lru->capacity = (size_t)std::atoi(e);
        if (debug_on()) std::fprintf(stderr, "[UGS INFO] Preprocessing cache initialized: size=%zu (set UGS_CACHE_SIZE to change)\n", g_lru->capacity);
    }
    return *g_lru;
}

uint64_t graph_key(const int64_t *u, const int64_t *v, int64_t cols, int64_t n) {
    const uint64_t prime = 1099511628211ull;
    uint64_t h = 14695981039346656037ull;
    h = (h ^ (uint64_t)n) * prime;
    h = (h ^ (uint64_t)cols) * prime;
    const int64_t step = cols > 1000 ? cols / 500 : 1;
    for (int64_t j = 0; j < cols; j += step) { h = (h ^ (uint64_t)u[j]) * prime; h = (h ^ (uint64_t)v[j]) * prime; }
    return h;
}

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
struct DeviceCtx { int id = -1; int cus = 256; hipStream_t stream = nullptr; };
std::mutex g_dev_mu;
std::map<int, DeviceCtx> &g_devs = *new std::map<int, DeviceCtx>();
thread_local int t_device = -1;
thread_local hipStream_t t_job_stream = nullptr;    // ugs_set_stream: stream of the calling thread's jobs (NULL: the library's own)
thread_local bool t_job_stream_set = false;

int device_ctx(DeviceCtx &out) {
    int dev = t_device;
    if (dev < 0) {
        int cnt = 0;
        hipError_t e = hipGetDeviceCount(&cnt);
        if (e != hipSuccess || cnt <= 0) return fail(UGS_E_NO_DEVICE, "no usable HIP device: this sampler has no CPU path (hipGetDeviceCount: " + std::string(hipGetErrorString(e)) + ")");
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    HIP_TRY(hipSetDevice(dev));
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = g_devs.find(dev);
    if (it == g_devs.end()) {
        DeviceCtx c;
        c.id = dev;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        it = g_devs.emplace(dev, c).first;
    }
    out = it->second;
    if (t_job_stream_set) out.stream = t_job_stream;
    return UGS_OK;
}

}  // namespace
int ugs_internal_ctx(int *device, hipStream_t *stream) {
    DeviceCtx dc;
    if (int rc = device_ctx(dc)) return rc;
    if (device) *device = dc.id;
    if (stream) *stream = dc.stream;
    return UGS_OK;
}
namespace {

// grow-only device scratch pool (per process): avoids hipMalloc/hipFree on every call
struct PoolBuf { void *p = nullptr; size_t bytes = 0; int dev = -1; bool owned = true; /* false: a piece of a slab, never hipFree'd on its own */ };
std::mutex g_pool_mu;
std::vector<PoolBuf> &g_pool_free = *new std::vector<PoolBuf>();
int pool_get(size_t bytes, int dev, PoolBuf &out) {
    if (bytes < 256) bytes = 256;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        int best = -1;
        for (int i = 0; i < (int)g_pool_free.size(); ++i)
            if (g_pool_free[(size_t)i].dev == dev && g_pool_free[(size_t)i].bytes >= bytes &&
                (best < 0 || g_pool_free[(size_t)i].bytes < g_pool_free[(size_t)best].bytes)) best = i;
        if (best >= 0 && g_pool_free[(size_t)best].bytes <= 4 * bytes + (1u << 20)) {
            out = g_pool_free[(size_t)best];
            g_pool_free.erase(g_pool_free.begin() + best);
            return UGS_OK;
        }
    }
    // Small requests (the plan of a mini-batch, per-call scratch) come in a slab of 16 one-megabyte pieces: a fresh hipMalloc per new plan
    // costs tens of microseconds -- up to 0.1 ms with a multi-gigabyte plan resident -- and the plan cache hands buffers back only once
    // its 64 slots are full, so without this the first 64 new mini-batches of a process each pay it.
    if (bytes <= ((size_t)1 << 20)) {
        const size_t piece = bytes <= ((size_t)64 << 10) ? (size_t)64 << 10 : (size_t)1 << 20;    // two classes: counters and lists / plans
        const int n = piece == ((size_t)1 << 20) ? 16 : 64;
        void *slab = nullptr;
        if (hipMalloc(&slab, (size_t)n * piece) == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            for (int i = 1; i < n; ++i) g_pool_free.push_back(PoolBuf{static_cast<char *>(slab) + (size_t)i * piece, piece, dev, false});
            out.p = slab; out.bytes = piece; out.dev = dev; out.owned = false;       // (the slab itself is never freed: 4 or 16 MB)
            return UGS_OK;
        }
        (void)hipGetLastError();
    }
    size_t rounded = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, rounded);
    if (e != hipSuccess) {   // drop the pool and retry once
        std::vector<PoolBuf> victims;
        { std::lock_guard<std::mutex> lk(g_pool_mu); victims.swap(g_pool_free); }
        for (auto &b : victims) if (b.owned) (void)hipFree(b.p);      // (pieces of a slab are dropped, their slab stays)
        e = hipMalloc(&p, rounded);
        if (e != hipSuccess) return fail_hip(e, "hipMalloc");
    }
    out.p = p; out.bytes = rounded; out.dev = dev; out.owned = true;
    return UGS_OK;
}
void pool_put(PoolBuf &b) {
    if (!b.p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_free.push_back(b);
    b = PoolBuf();
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------------------------------------------
struct TierChoice { int first = UGS_TIER_S; int second = -1 /* LDS tier that redoes the rows the first one hands on */; bool third_G = false; int64_t bound = 0;
                    bool small = false /* tier S in its 32-candidate form */; bool wide = false /* tier S may run with 16 lanes per walk */; };

struct ugs_plan {
    int device = -1, cus = 256;
    int64_t G = 0, nverts = 0, nnz = 0;
    void *blob = nullptr;                 // one device allocation holding every array of the plan
    size_t blob_bytes = 0;
    PoolBuf blob_buf;                     // small plans borrow their allocation from the scratch pool
    UgsPlanDev dev{};
    // per-graph statistics used to pick the walk tier for a given k
    std::vector<int64_t> g_n, g_maxdeg;
    std::vector<double> g_sbdeg;
    std::vector<int> g_level;
    std::mutex mu;
    std::map<int, TierChoice> tiers;
    std::atomic<int> refs{1};
    uint64_t cache_key = 0;
    bool cached = false;
    // lazily grown scratch owned by the plan (serialised by `mu` per call)
    PoolBuf counts, ovf1, ovf2, ovfcnt, scantmp, gws;
    // edges staged by the last walk (UgsWalkArgs::stage) and the call they belong to: a fill of exactly those rows into/from
    // the same nodes buffer expands them; any other fill reads the adjacency rows again
    PoolBuf stage, ulist, work;   // work: 3 x u64 next-item counters (one per walk launch of a call)
    // scan folded into the fill (ugs_plan_step, ugs_fill_scan): the walk kernel's sums of 8 consecutive rows
    PoolBuf tiles;
    // job path: the scan kernel hands the edge total to the host through 16 bytes of pinned memory (total, epoch word) and the host
    // polls the word instead of copying the total back behind a stream wait (wait_signal)
    char *pin_slot = nullptr;
    uint32_t pin_epoch = 0;
    // stream order between calls: a plan's scratch is reused by every call, so a call on another stream than the previous one
    // first waits (on the device) for that call's last kernel
    hipEvent_t last_ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool last_valid = false;
    PoolBuf prow;                         // padded rows (ugs_device.h), built on the device by the first walk in a one-walk-per-wave tier
    std::atomic<size_t> prow_bytes{0};    // their size, readable by the plan cache without this plan's mutex
    bool prow_pooled = false, prow_failed = false;
    int walk_share = 100;                 // ugs_plan_set_walk_share
    bool stg_valid = false;
    const void *stg_nodes = nullptr;
    int64_t stg_row_begin = 0, stg_row_count = 0;
    int stg_m = 0, stg_k = 0;
    int64_t gws_groups = 0, gws_words = 0;
    int gcap = 0, ghs = 0, gbcap = 0, gpcap = 0;
    std::vector<std::shared_ptr<void>> keep;   // device-built plans point into their graphs' arena entries: the graphs live as long as the plan
    ugs_plan *twin_of = nullptr;               // ugs_plan_twin: the plan whose device arrays this one shares (it holds a reference)
    UgsLaunchInfo last_walk{nullptr, 0, 0, 0};
    UgsLaunchInfo last_fill{nullptr, 0, 0, 0};
    int64_t last_overflow = 0;
    bool handle_api = false;
    // optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline figure)
    bool timing = false;
    struct EvPair { hipEvent_t a, b; int kind; };     // kind 0 = first-tier walk kernel, 1 = overflow tiers + scan, 2 = fill kernel
    std::vector<EvPair> events;
};

namespace {

struct PlanPiece {                 // one graph of a plan
    std::shared_ptr<Graph> g;      // null for degenerate graphs
    int64_t lo = 0;
    const int64_t *colmap = nullptr;   // batch column of each of the graph's columns (null: identity)
    int64_t ncols = 0;                 // entries of colmap
};

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Large graphs: the O(nnz) preprocessing stages on the GPU (ugs_preproc.hip).  The degree order stays on
// the host (O(n)); the outcome is bit-identical to the host path.  UGS_DEVICE_PREPROC=0 never, =1 always (testing aid);
// by default graphs with >= 2^21 columns when a device is present and has room.  `done` = false -> caller runs the host path.
int preprocess_on_device(Graph &G, const int64_t *src, const int64_t *dst, int64_t E, int k, bool &done) {
    done = false;
    const char *env = std::getenv("UGS_DEVICE_PREPROC");
    if (env && env[0] == '0') return UGS_OK;
    const bool forced = env && env[0] == '1';
    if (G.n < 1 || E < 1 || E >= ((int64_t)1 << 30)) return UGS_OK;
    if (!forced && E < ((int64_t)1 << 21)) return UGS_OK;
    DeviceCtx dc;
    {
        const std::string keep = t_err;
        if (device_ctx(dc) != UGS_OK) { t_err = keep; return UGS_OK; }       // no device: create_preproc still works on the host
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || ugs_devpre_bytes(G.n, E) > free_b / 2) return UGS_OK;
    UgsDevPre *d = nullptr;
    struct Guard { UgsDevPre *&d; ~Guard() { ugs_devpre_free(d); } } guard{d};
    Lap lap;
    G.rowptr.assign((size_t)G.n + 1, 0);
    HIP_TRY(ugs_devpre_csr(&d, src, dst, E, G.n, dc.stream, G.rowptr.data(), &G.nnz));
    if (G.nnz >= (int64_t)INT32_MAX) return fail(UGS_E_UNSUPPORTED, "graph too large: CSR entries must be < 2^31 - 1");
    const double t_csr = lap();
    order_by_degree(G);
    const double t_order = lap();
    G.sdeg.assign((size_t)G.n, 0);
    G.reach.assign((size_t)G.n, 0);
    HIP_TRY(ugs_devpre_roots(d, G.order.data(), G.rank.data(), k, G.sdeg.data(), G.reach.data()));
    const double t_roots = lap();
    G.nbr.resize((size_t)G.nnz);
    G.col.resize((size_t)G.nnz);
    HIP_TRY(ugs_devpre_download(d, G.nbr.data(), G.col.data()));
    if (debug_on())
        std::fprintf(stderr, "[UGS PREPROC] n=%lld columns=%lld on the device: upload+CSR %.3fs, degree order (host) %.3fs, suffix degrees+reachability %.3fs, "
                     "CSR download %.3fs\n", (long long)G.n, (long long)E, t_csr, t_order, t_roots, lap());
    ugs_devpre_trim(d);
    { std::lock_guard<std::mutex> lk(G.pre_mu); G.devpre = d; G.devpre_dev = dc.id; d = nullptr; }
    done = true;
    return UGS_OK;
}

int assemble_plan(const std::vector<PlanPiece> &pieces, const DeviceCtx &dc, ugs_plan *plan) {
    const int64_t G = (int64_t)pieces.size();
    int64_t nv = 0, nnz = 0, nrows = 0, nviable = 0;
    for (auto &pc : pieces) if (pc.g) { nv += pc.g->n; nnz += pc.g->nnz; nrows += pc.g->n + 1; if (pc.g->level > 0) nviable += (int64_t)pc.g->viable.size(); }
    if (nnz >= (int64_t)INT32_MAX) return fail(UGS_E_UNSUPPORTED, "plan too large: total CSR entries must be < 2^31 - 1");
    size_t off_desc = 0;
    size_t off_row = align_up(off_desc + (size_t)std::max<int64_t>(G, 1) * sizeof(UgsGraphDesc));
    size_t off_adj = align_up(off_row + (size_t)std::max<int64_t>(nrows, 1) * sizeof(int64_t));
    size_t off_col = align_up(off_adj + (size_t)std::max<int64_t>(nnz, 1) * sizeof(int2));
    size_t off_root = align_up(off_col + (size_t)std::max<int64_t>(nnz, 1) * sizeof(int2));
    size_t off_via = align_up(off_root + (size_t)std::max<int64_t>(nv, 1) * sizeof(UgsRootRec));
    size_t total = align_up(off_via + (size_t)std::max<int64_t>(nviable, 1) * sizeof(int2));
    // Graphs preprocessed on this device still hold their CSR in HBM (Graph::devpre): their adjacency arrays are written by
    // a kernel and never cross the bus.  Without such a piece (every batch of small graphs) the plan is ONE host blob, one copy.
    std::vector<UgsDevPre *> dpre((size_t)G, nullptr);
    bool split = false;
    for (int64_t gi = 0; gi < G; ++gi)
        if (pieces[(size_t)gi].g && (dpre[(size_t)gi] = pieces[(size_t)gi].g->take_devpre(dc.id))) split = true;
    struct Owned { std::vector<UgsDevPre *> &v; ~Owned() { for (auto *d : v) ugs_devpre_free(d); } } owned{dpre};
    std::vector<char> host(split ? off_adj : total, 0);           // split: head = descriptors + row pointer
    std::vector<char> tail(split ? total - off_root : 0, 0);      // split: tail = root records + viable lists
    auto *desc = reinterpret_cast<UgsGraphDesc *>(host.data() + off_desc);
    auto *rowp = reinterpret_cast<int64_t *>(host.data() + off_row);
    auto *adj = split ? nullptr : reinterpret_cast<int2 *>(host.data() + off_adj);
    auto *adjf = split ? nullptr : reinterpret_cast<int2 *>(host.data() + off_col);
    char *tail_base = split ? tail.data() - off_root : host.data();
    auto *roots = reinterpret_cast<UgsRootRec *>(tail_base + off_root);
    auto *via = reinterpret_cast<int2 *>(tail_base + off_via);
    void *dptr = nullptr;
    if (total <= ((size_t)64 << 20)) {      // batches of small graphs come and go with every shuffled mini-batch: pooled
        if (int rc = pool_get(total, dc.id, plan->blob_buf)) return rc;
        dptr = plan->blob_buf.p;
    } else {
        HIP_TRY(hipMalloc(&dptr, total));
    }
    auto give_back = [&] { if (plan->blob_buf.p) { pool_put(plan->blob_buf); plan->blob_buf = PoolBuf(); } else (void)hipFree(dptr); };
    char *base = static_cast<char *>(dptr);
    auto fill_adj = [](const PlanPiece &pc, int2 *a, int2 *af) {
        const Graph &g = *pc.g;
        for (int64_t p = 0; p < g.nnz; ++p) {
            const int32_t w = g.nbr[(size_t)p];
            a[p] = make_int2(w, g.rank[(size_t)w]);
            const int32_t c = g.col[(size_t)p];
            af[p] = make_int2(w, pc.colmap ? (int32_t)pc.colmap[c] : c);
        }
    };
    int64_t rb = 0, vb = 0, ab = 0, vib = 0;
    plan->g_n.resize((size_t)G); plan->g_maxdeg.resize((size_t)G); plan->g_sbdeg.resize((size_t)G); plan->g_level.resize((size_t)G);
    std::vector<int2> tmp_a, tmp_f;
    for (int64_t gi = 0; gi < G; ++gi) {
        const PlanPiece &pc = pieces[(size_t)gi];
        UgsGraphDesc &d = desc[gi];
        d.node_lo = pc.lo; d.rbase = rb; d.vbase = vb; d.viable_base = vib; d.pad = 0;
        if (!pc.g) { d.n = 0; d.level = -1; d.n_viable = 0; plan->g_n[(size_t)gi] = 0; plan->g_maxdeg[(size_t)gi] = 0; plan->g_sbdeg[(size_t)gi] = 0; plan->g_level[(size_t)gi] = -1; continue; }
        const Graph &g = *pc.g;
        d.n = (int32_t)g.n; d.level = g.level; d.n_viable = (int32_t)g.viable.size();
        plan->g_n[(size_t)gi] = g.n; plan->g_maxdeg[(size_t)gi] = g.max_deg; plan->g_sbdeg[(size_t)gi] = g.sb_deg; plan->g_level[(size_t)gi] = g.level;
        for (int64_t r = 0; r <= g.n; ++r) rowp[rb + r] = ab + g.rowptr[(size_t)r];
        if (!split) fill_adj(pc, adj + ab, adjf + ab);
        else {
            int2 *d_adj = reinterpret_cast<int2 *>(base + off_adj) + ab, *d_adjf = reinterpret_cast<int2 *>(base + off_col) + ab;
            hipError_t e = hipSuccess;
            if (UgsDevPre *dp = dpre[(size_t)gi]) {
                const int64_t *cm = pc.colmap;
                if (cm) { bool ident = true; for (int64_t c = 0; c < pc.ncols && ident; ++c) ident = cm[c] == c; if (ident) cm = nullptr; }
                e = ugs_devpre_assemble(dp, cm, pc.ncols, d_adj, d_adjf, dc.stream);
            } else if (g.nnz > 0) {
                tmp_a.resize((size_t)g.nnz); tmp_f.resize((size_t)g.nnz);
                fill_adj(pc, tmp_a.data(), tmp_f.data());
                e = hipMemcpy(d_adj, tmp_a.data(), (size_t)g.nnz * sizeof(int2), hipMemcpyHostToDevice);
                if (e == hipSuccess) e = hipMemcpy(d_adjf, tmp_f.data(), (size_t)g.nnz * sizeof(int2), hipMemcpyHostToDevice);
            }
            if (e != hipSuccess) { give_back(); return fail_hip(e, "plan adjacency"); }
        }
        if (g.level == 0)
            for (int64_t vi = 0; vi < g.n; ++vi) {
                UgsRootRec &r = roots[vb + vi];
                r.prob = g.prob[(size_t)vi]; r.alias = g.alias[(size_t)vi];
                r.v_self = g.order[(size_t)vi]; r.v_alias = g.order[(size_t)g.alias[(size_t)vi]]; r.pad = 0;
            }
        else
            for (size_t t = 0; t < g.viable.size(); ++t) via[vib + (int64_t)t] = make_int2(g.viable[t], g.order[(size_t)g.viable[t]]);
        rb += g.n + 1; vb += g.n; ab += g.nnz; if (g.level > 0) vib += (int64_t)g.viable.size();
    }
    hipError_t e = hipMemcpy(dptr, host.data(), host.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && split) e = hipMemcpy(base + off_root, tail.data(), tail.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { give_back(); return fail_hip(e, "hipMemcpy(plan)"); }
    plan->device = dc.id; plan->cus = dc.cus;
    plan->G = G; plan->nverts = nv; plan->nnz = nnz;
    plan->blob = dptr; plan->blob_bytes = total;
    plan->dev.graphs = reinterpret_cast<const UgsGraphDesc *>(base + off_desc);
    plan->dev.rowptr = reinterpret_cast<const int64_t *>(base + off_row);
    plan->dev.adj = reinterpret_cast<const int2 *>(base + off_adj);
    plan->dev.adjf = reinterpret_cast<const int2 *>(base + off_col);
    plan->dev.roots = reinterpret_cast<const UgsRootRec *>(base + off_root);
    plan->dev.viable = reinterpret_cast<const int2 *>(base + off_via);
    plan->dev.num_graphs = G;
    return UGS_OK;
}

hipError_t ev_begin(ugs_plan *p, int kind, hipStream_t s) {
    if (!p->timing) return hipSuccess;
    ugs_plan::EvPair e{nullptr, nullptr, kind};
    hipError_t r = hipEventCreate(&e.a);
    if (r == hipSuccess) r = hipEventCreate(&e.b);
    if (r == hipSuccess) r = hipEventRecord(e.a, s);
    p->events.push_back(e);
    return r;
}
hipError_t ev_end(ugs_plan *p, hipStream_t s) {
    if (!p->timing || p->events.empty()) return hipSuccess;
    return hipEventRecord(p->events.back().b, s);
}
void ev_clear(ugs_plan *p) {
    for (auto &e : p->events) { if (e.a) (void)hipEventDestroy(e.a); if (e.b) (void)hipEventDestroy(e.b); }
    p->events.clear();
}

void destroy_plan(ugs_plan *p) {
    if (!p) return;
    if (p->device >= 0 && hipSetDevice(p->device) == hipSuccess) (void)hipDeviceSynchronize();   // kernels may still read the plan
    ev_clear(p);
    if (p->last_ev) (void)hipEventDestroy(p->last_ev);
    if (p->blob_buf.p) pool_put(p->blob_buf); else if (p->blob) (void)hipFree(p->blob);
    pool_put(p->counts); pool_put(p->ovf1); pool_put(p->ovf2); pool_put(p->ovfcnt); pool_put(p->scantmp);
    pool_put(p->stage); pool_put(p->ulist); pool_put(p->work); pool_put(p->tiles);
    pin_slot_put(p->pin_slot);
    if (p->prow.p) { if (p->prow_pooled) pool_put(p->prow); else (void)hipFree(p->prow.p); p->prow = PoolBuf(); }
    if (p->gws.p) { (void)hipFree(p->gws.p); p->gws = PoolBuf(); }
    ugs_plan *owner = p->twin_of;
    delete p;
    if (owner && owner->refs.fetch_sub(1) == 1) destroy_plan(owner);
}
}  // namespace
namespace { void plan_unref(ugs_plan *p) { if (p && p->refs.fetch_sub(1) == 1) destroy_plan(p); } }
namespace {

// plan cache of batches (small LRU; a hit skips assembly + upload)
std::mutex g_pc_mu;
std::list<ugs_plan *> &g_plan_cache = *new std::list<ugs_plan *>();     // front = most recent
size_t g_plan_cache_cap = 64;
size_t g_plan_cache_bytes_cap = (size_t)8 << 30;

ugs_plan *plan_cache_get(uint64_t key, int dev) {
    std::lock_guard<std::mutex> lk(g_pc_mu);
    for (auto it = g_plan_cache.begin(); it != g_plan_cache.end(); ++it)
        if ((*it)->cache_key == key && (*it)->device == dev) {
            ugs_plan *p = *it;
            g_plan_cache.splice(g_plan_cache.begin(), g_plan_cache, it);
            p->refs.fetch_add(1);
            return p;
        }
    return nullptr;
}
void plan_cache_put(ugs_plan *p) {
    std::vector<ugs_plan *> victims;
    {
        std::lock_guard<std::mutex> lk(g_pc_mu);
        p->refs.fetch_add(1);
        p->cached = true;
        g_plan_cache.push_front(p);
        size_t bytes = 0;
        for (auto *q : g_plan_cache) bytes += q->blob_bytes + q->prow_bytes.load();
        while (g_plan_cache.size() > g_plan_cache_cap || (bytes > g_plan_cache_bytes_cap && g_plan_cache.size() > 1)) {
            ugs_plan *v = g_plan_cache.back();
            g_plan_cache.pop_back();
            bytes -= v->blob_bytes + v->prow_bytes.load();
            victims.push_back(v);
        }
    }
    for (auto *v : victims) plan_unref(v);
}
void plan_cache_trim() {      // after a cached plan grew (padded rows built lazily): apply the byte cap again
    std::vector<ugs_plan *> victims;
    {
        std::lock_guard<std::mutex> lk(g_pc_mu);
        size_t bytes = 0;
        for (auto *q : g_plan_cache) bytes += q->blob_bytes + q->prow_bytes.load();
        while (bytes > g_plan_cache_bytes_cap && g_plan_cache.size() > 1) {
            ugs_plan *v = g_plan_cache.back();
            g_plan_cache.pop_back();
            bytes -= v->blob_bytes + v->prow_bytes.load();
            victims.push_back(v);
        }
    }
    for (auto *v : victims) plan_unref(v);
}
void plan_cache_clear() {
    std::vector<ugs_plan *> victims;
    { std::lock_guard<std::mutex> lk(g_pc_mu); victims.assign(g_plan_cache.begin(), g_plan_cache.end()); g_plan_cache.clear(); }
    for (auto *v : victims) plan_unref(v);
}

// ---------------------------------------------------------------------------------------------------------------
// Repeated batches.  The reference interface hands the whole batch over on every call, and the reference (like the general
// path below) walks every column of it to slice, renumber and hash the graphs before it finds its preprocessing in the LRU.
// A training loop calls with the SAME batch tensors again and again (gps/experiment.py:882-883, fixed seed), so calls are
// first matched as a whole: a 128-bit content hash of (edge_index rows, ptr) -- two independent 64-bit multiply-fold lanes
// over every byte, one pass at memory speed -- names the batch; a hit replays exactly what the general path would do to the
// LRU (one lookup per graph, in graph order: same recency order, same hit counters) and, if every graph is still cached under
// the same handle and the plan is still in the plan cache, returns that plan.  Anything else falls through to the general
// path.  Observable behaviour is that of the general path; the cost drops from several passes over the columns with
// per-column searches and copies (0.24 s per call on the 20M-column graph) to one hashing pass.
// ---------------------------------------------------------------------------------------------------------------
struct Hash128 { uint64_t a, b; bool operator==(const Hash128 &o) const { return a == o.a && b == o.b; } };
inline uint64_t mum(uint64_t x, uint64_t y) { const __uint128_t r = (__uint128_t)x * y; return (uint64_t)r ^ (uint64_t)(r >> 64); }
void hash_words(const int64_t *p, int64_t n, Hash128 &h) {
    const uint64_t K0 = 0xa0761d6478bd642full, K1 = 0xe7037ed1a0b428dbull, K2 = 0x8ebc6af09c88c6e3ull, K3 = 0x589965cc75374cc3ull;
    uint64_t a0 = h.a, a1 = h.a ^ K2, b0 = h.b, b1 = h.b ^ K3;
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {      // two accumulators per lane: the multiplies of consecutive groups overlap
        const uint64_t x0 = (uint64_t)p[i], x1 = (uint64_t)p[i + 1], x2 = (uint64_t)p[i + 2], x3 = (uint64_t)p[i + 3];
        a0 = mum(x0 ^ K0, x1 ^ a0); a1 = mum(x2 ^ K0, x3 ^ a1);
        b0 = mum(x1 ^ K1, x0 ^ b0); b1 = mum(x3 ^ K1, x2 ^ b1);
    }
    for (; i < n; ++i) { a0 = mum((uint64_t)p[i] ^ K0, a0 ^ K2); b0 = mum((uint64_t)p[i] ^ K1, b0 ^ K3); }
    h.a = mum(a0 ^ K1, a1 ^ (uint64_t)n); h.b = mum(b0 ^ K0, b1 ^ (uint64_t)n);
}
// Equality of two batches is decided by their 128-bit content hash -- probabilistic, and the multiply-fold lanes are not
// collision-resistant against crafted input -- plus a WITNESS of 32 words read back from fixed places of the batch (first, last and
// evenly strided columns of both rows and the ends of ptr): a false match would need a hash collision between two batches that also
// agree in those words.
struct Witness {
    int64_t w[32];
    bool operator==(const Witness &o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
};
Witness batch_witness(const int64_t *src, const int64_t *dst, int64_t E, const int64_t *ptr, int64_t G) {
    Witness x{};
    for (int i = 0; i < 14; ++i) {
        const int64_t j = E > 0 ? (int64_t)((__int128)i * (E - 1) / 13) : 0;
        x.w[i] = E > 0 ? src[j] : 0;
        x.w[14 + i] = E > 0 ? dst[j] : 0;
    }
    x.w[28] = ptr[0]; x.w[29] = ptr[G]; x.w[30] = ptr[G / 2]; x.w[31] = E ^ (G << 40);
    return x;
}
struct BatchEntry {
    Hash128 h; int64_t E, G; int k, dev;
    uint64_t plan_key;
    std::vector<std::pair<uint64_t, int64_t>> graphs;     // (LRU key, handle) of every non-degenerate graph, in graph order
    Witness wit;
};
std::mutex g_bi_mu;
std::list<BatchEntry> &g_batch_index = *new std::list<BatchEntry>();   // front = most recent, at most as many as cached plans
void batch_index_clear() { std::lock_guard<std::mutex> lk(g_bi_mu); g_batch_index.clear(); }
// The plan a batch PROBABLY maps to, from the 32 sampled words alone (no pass over the columns, no LRU effect): the streamed call
// starts its walks on it while the real lookup -- content hash, LRU replay -- runs, and keeps them only if that lookup names the same plan.
ugs_plan *peek_batch_plan(const int64_t *src, const int64_t *dst, int64_t E, const int64_t *ptr, int64_t G, int k, int dev) {
    const Witness wit = batch_witness(src, dst, E, ptr, G);
    uint64_t key = 0;
    {
        std::lock_guard<std::mutex> lk(g_bi_mu);
        auto it = g_batch_index.begin();
        for (; it != g_batch_index.end(); ++it)
            if (it->E == E && it->G == G && it->k == k && it->dev == dev && it->wit == wit) { key = it->plan_key; break; }
        if (it == g_batch_index.end()) return nullptr;
    }
    return plan_cache_get(key, dev);
}

// A long array is hashed in chunks of a FIXED size (the value must not depend on the number of threads) by a few helper
// threads; the chunk hashes are then hashed in order.  (20M columns: the hashing pass drops from ~10 ms to ~2 ms of a 20 ms call.)
void hash_array(const int64_t *p, int64_t n, Hash128 &h) {
    constexpr int64_t CH = (int64_t)1 << 20;                                  // words per chunk (8 MB)
    if (n < 4 * CH) { hash_words(p, n, h); return; }
    const int64_t nch = (n + CH - 1) / CH;
    std::vector<Hash128> part((size_t)nch);
    const Hash128 seed = h;
    auto work = [&](int64_t first, int64_t stride) {
        for (int64_t c = first; c < nch; c += stride) {
            Hash128 x{seed.a ^ (0x9e3779b97f4a7c15ull * (uint64_t)(c + 1)), seed.b + (uint64_t)c};
            hash_words(p + c * CH, std::min<int64_t>(CH, n - c * CH), x);
            part[(size_t)c] = x;
        }
    };
    const int64_t hw = (int64_t)std::thread::hardware_concurrency();
    const int64_t T = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nch, 8), hw > 0 ? hw : 1));
    std::vector<std::thread> helpers;
    int64_t started = 1;                                                       // strides taken by a thread: 0 (this one), 1 .. started-1
    try {
        for (int64_t t = 1; t < T; ++t) { helpers.emplace_back(work, t, T); started = t + 1; }
    } catch (...) {}                                                           // std::system_error must not cross the C ABI: this thread does the rest
    work(0, T);
    for (int64_t t = started; t < T; ++t) work(t, T);
    for (auto &th : helpers) th.join();
    static_assert(sizeof(Hash128) == 2 * sizeof(int64_t), "chunk hashes are hashed as words");
    hash_words(reinterpret_cast<const int64_t *>(part.data()), 2 * nch, h);
}

Hash128 batch_hash(const int64_t *src, const int64_t *dst, int64_t E, const int64_t *ptr, int64_t G) {
    Hash128 h{0x2d358dccaa6c78a5ull ^ (uint64_t)E, 0x8bb84b93962eacc9ull ^ (uint64_t)G};
    hash_words(ptr, G + 1, h);
    hash_array(src, E, h);
    hash_array(dst, E, h);
    return h;
}

// the LRU touches of a matched batch; false = some graph is no longer cached under the recorded handle (caller takes the general path)
bool batch_replay(const BatchEntry &e, int64_t &hits) {
    std::lock_guard<std::mutex> lk(g_lru_mu);
    for (auto &kh : e.graphs) {
        int64_t handle = 0;
        if (!lru().get(kh.first, handle) || handle != kh.second) return false;
        ++lru().hits; ++hits;
    }
    return true;
}

// pick the walk tier(s) for k: the candidate set of a walk holds at most min(n-1, (k-1)*max degree) vertices
TierChoice choose_tier(ugs_plan *p, int k) {
    std::lock_guard<std::mutex> lk(p->mu);
    auto it = p->tiers.find(k);
    if (it != p->tiers.end()) return it->second;
    int64_t bound = 0;
    double mean = 0;            // expected candidates at the last step: (k-1) rows of size-biased mean degree
    for (size_t gi = 0; gi < p->g_n.size(); ++gi) {
        if (p->g_level[gi] < 0) continue;
        const int64_t b = std::max<int64_t>(0, std::min<int64_t>(p->g_n[gi] - 1, (int64_t)(k - 1) * p->g_maxdeg[gi]));
        bound = std::max(bound, b);
        mean = std::max(mean, std::min<double>((double)b, (double)(k - 1) * p->g_sbdeg[gi]));
    }
    TierChoice t;
    t.bound = bound;
    // First tier.  8 lanes per walk only with a safety margin (a handed-on walk of a small graph is redone with a whole wave).
    // Among the 64-lane tiers the smaller one pays as long as most walks fit: measured on ER degree 40, k = 12 (mean 451),
    // starting in the 448 tier hands 1.4 % of the rows on and runs at 47.8 M/s against 33.4 M/s when everything starts in
    // the 1024 tier; k = 10 (mean 369): 68 against 43 M/s.
    t.first = UGS_TIER_L;
    if (1.3 * mean + 16.0 <= UGS_TIER_CAP[UGS_TIER_S] || bound <= UGS_TIER_CAP[UGS_TIER_S]) t.first = UGS_TIER_S;
    else
        for (int tier = UGS_TIER_M; tier < UGS_TIER_L; ++tier)
            if (mean <= 1.05 * UGS_TIER_CAP[tier] || bound <= UGS_TIER_CAP[tier]) { t.first = tier; break; }
    if (const char *e = std::getenv("UGS_FORCE_TIER")) { int f = std::atoi(e); if (f >= 0 && f < UGS_LDS_TIERS) t.first = f; }
    // A tier hands a walk on when its candidate list would overflow (bound > CAP) OR when its hash-table guard trips:
    // `vertices seen + candidate lanes of the chunk > hash limit`, which is conservative (the lanes need not be new).  A walk
    // has seen at most bound + 1 vertices, so the guard can trip as soon as bound + 1 + lanes-per-walk exceeds the limit --
    // for the 448-candidate tier (limit 448, 64 lanes) already from bound 384.
    auto may_hand_on = [&](int tier) { return bound > UGS_TIER_CAP[tier] || bound + 1 + UGS_TIER_LANES[tier] > UGS_TIER_HASH_LIMIT[tier]; };
    // handed-on rows are redone ONCE by the smallest LDS tier that cannot hand on itself (else the largest), then by the
    // global-memory tier if even that one can
    if (t.first < UGS_TIER_L && may_hand_on(t.first)) {
        t.second = UGS_TIER_L;
        for (int tier = t.first + 1; tier < UGS_TIER_L; ++tier) if (!may_hand_on(tier)) { t.second = tier; break; }
    }
    const int last_lds = t.second >= 0 ? t.second : t.first;
    t.third_G = may_hand_on(last_lds);
    // tier S in its 32-candidate form: only where no walk can outgrow it (candidate bound, and the hash guard: seen vertices + the
    // 8 lanes of a chunk), so that it needs no hand-on chain of its own
    t.small = t.first == UGS_TIER_S && bound <= UGS_SMALL_CAP && bound + 1 + UGS_TIER_LANES[UGS_TIER_S] <= UGS_SMALL_HASH_LIMIT &&
              std::getenv("UGS_NO_SMALL_TIER") == nullptr;
    // tier S with 16 lanes per walk (UGS_WIDE_LANES; plan_walk_impl takes it for launches that fit the GPU at once): only where no walk
    // can be handed on with 16-lane chunks either
    t.wide = t.first == UGS_TIER_S && (t.small ? bound + 1 + UGS_WIDE_LANES <= UGS_SMALL_HASH_LIMIT
                                               : bound <= UGS_TIER_CAP[UGS_TIER_S] && bound + 1 + UGS_WIDE_LANES <= UGS_TIER_HASH_LIMIT[UGS_TIER_S]);
    p->tiers[k] = t;
    return t;
}

// Column j belongs to graph g iff both endpoints lie in [ptr[g], ptr[g+1]) (the reference lets every graph scan every
// column, src/ugs_sampler_batch_extension.cpp:41-75; with a monotone ptr the node ranges are disjoint and one pass with a
// binary search gives the same per-graph lists, in column order).
void assign_columns(const int64_t *src, const int64_t *dst, int64_t E, const int64_t *ptr, int64_t G,
                    std::vector<int64_t> &cstart, std::vector<int64_t> &cols_of) {
    cstart.assign((size_t)G + 1, 0);
    cols_of.clear();
    bool monotone = true;
    for (int64_t g = 0; g < G; ++g) if (ptr[g + 1] < ptr[g]) { monotone = false; break; }
    if (monotone) {
        std::vector<int32_t> owner((size_t)E, -1);
        for (int64_t j = 0; j < E; ++j) {
            const int64_t u = src[j], v = dst[j];
            if (G == 0 || u < ptr[0] || u >= ptr[G]) continue;
            const int64_t g = (std::upper_bound(ptr, ptr + G + 1, u) - ptr) - 1;    // last g with ptr[g] <= u
            if (g < 0 || g >= G || !(u >= ptr[g] && u < ptr[g + 1])) continue;
            if (v >= ptr[g] && v < ptr[g + 1]) { owner[(size_t)j] = (int32_t)g; ++cstart[(size_t)g + 1]; }
        }
        for (int64_t g = 0; g < G; ++g) cstart[(size_t)g + 1] += cstart[(size_t)g];
        cols_of.resize((size_t)cstart[(size_t)G]);
        std::vector<int64_t> wr(cstart.begin(), cstart.end() - 1);
        for (int64_t j = 0; j < E; ++j) if (owner[(size_t)j] >= 0) cols_of[(size_t)wr[(size_t)owner[(size_t)j]]++] = j;
    } else {   // arbitrary ptr: every graph scans every column, like the reference
        for (int64_t g = 0; g < G; ++g) {
            const int64_t lo = ptr[g], hi = ptr[g + 1];
            for (int64_t j = 0; j < E; ++j) if (src[j] >= lo && src[j] < hi && dst[j] >= lo && dst[j] < hi) cols_of.push_back(j);
            cstart[(size_t)g + 1] = (int64_t)cols_of.size();
        }
    }
}

// Padded rows for the one-walk-per-wave tiers (ugs_device.h): 2^shift entries per vertex, sized so that the rows a walk visits
// (degree ~ the size-biased mean) fit their block with three standard deviations to spare; longer rows continue in adj[].
// Built by a kernel from the plan's own CSR, once.  UGS_NO_PROW=1 keeps the row-pointer path (A/B measurements, tests of both).
// Call with plan->mu held.  Failure to allocate is not an error: the walk then reads rows through the row pointer.
int ensure_prow(ugs_plan *plan, hipStream_t s) {
    if (plan->dev.prow || plan->prow_failed || plan->nverts <= 0) return UGS_OK;
    if (const char *e = std::getenv("UGS_NO_PROW")) if (e[0] == '1') return UGS_OK;
    double sb = 0;
    for (double x : plan->g_sbdeg) sb = std::max(sb, x);
    int shift = 3;
    while (shift < 6 && (double)((1 << shift) - 1) < sb + 3.0 * std::sqrt(sb) + 1.0) ++shift;
    if (const char *e = std::getenv("UGS_PROW_SHIFT")) { const int f = std::atoi(e); if (f >= 3 && f <= 6) shift = f; }
    const size_t bytes = ((size_t)plan->nverts << shift) * sizeof(int2);
    if (bytes >= ((size_t)1 << 32)) { plan->prow_failed = true; return UGS_OK; }       // the walk kernels address a block by a 32-bit byte offset
    if (bytes <= ((size_t)64 << 20)) {
        const std::string keep = t_err;
        if (pool_get(bytes, plan->device, plan->prow) != UGS_OK) { t_err = keep; plan->prow_failed = true; return UGS_OK; }
        plan->prow_pooled = true;
    } else {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); plan->prow_failed = true; return UGS_OK; }
        plan->prow.p = p; plan->prow.bytes = bytes; plan->prow.dev = plan->device;
    }
    HIP_TRY(ugs_launch_build_prow(plan->dev, plan->nverts, static_cast<int2 *>(plan->prow.p), shift, plan->cus, s));
    HIP_TRY(hipStreamSynchronize(s));          // once per plan: later calls may come on other streams
    plan->dev.prow = static_cast<const int2 *>(plan->prow.p);
    plan->dev.prow_shift = shift;
    // entries fetched together with the header: whole 128-byte lines (16 entries) covering the typical visited row; a row that
    // reaches further costs one more (dependent) load of the block's remaining lines
    int first = (int)std::ceil((sb + 0.5 * std::sqrt(sb) + 1.0) / 16.0) * 16;
    if (const char *e = std::getenv("UGS_PROW_FIRST")) { const int f = std::atoi(e); if (f >= 1) first = f; }
    plan->dev.prow_first = std::max(1, std::min(first, 1 << shift));
    plan->prow_bytes.store(plan->prow.bytes);
    if (plan->cached) plan_cache_trim();        // the cache admitted the plan without these bytes: apply its byte cap again
    return UGS_OK;
}

// plan scratch, grown on demand.  The old buffer goes back to the process-wide pool, where any stream may pick it up: the
// plan's last call has to be over first (rare: sizes settle after the first calls).
int ensure(PoolBuf &b, size_t bytes, int dev, ugs_plan *plan) {
    if (b.p && b.bytes >= bytes) return UGS_OK;
    if (b.p && plan && plan->last_valid) HIP_TRY(hipEventSynchronize(plan->last_ev));     // the event, not the caller's stream handle (which may be gone)
    pool_put(b);
    return pool_get(bytes, dev, b);
}

bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
// call with plan->mu held, before the first / after the last kernel of a call on stream s
int plan_enter(ugs_plan *plan, hipStream_t s) {
    if (plan->last_valid && plan->last_stream != s && !capturing(s)) HIP_TRY(hipStreamWaitEvent(s, plan->last_ev, 0));
    return UGS_OK;
}
int plan_leave(ugs_plan *plan, hipStream_t s) {
    if (capturing(s)) return UGS_OK;
    if (!plan->last_ev) HIP_TRY(hipEventCreateWithFlags(&plan->last_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(plan->last_ev, s));
    plan->last_stream = s; plan->last_valid = true;
    return UGS_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Device batch pass (batches of small graphs; kernels: ugs_batch.hip).
//
// A call with a batch that was not seen before as a whole -- in a training loop: every shuffled mini-batch, a new combination
// of graphs the LRU already knows -- no longer walks the columns on the host.  edge_index and ptr are uploaded, one device pass
// slices the batch, computes every graph's LRU key (the reference's FNV-1a, include/cache.hpp:81-109) and writes the plan's
// rowptr / adj / adjf straight into HBM in the reference's CSR order (src/preproc.cpp:32-86); 16 bytes per graph come back.
// The host replays the LRU on those keys (same lookups, same order, same eviction as the general path) and points every
// graph's descriptor at the root records its cached preprocessing keeps in the device's arena.  Only a graph the LRU does not
// know is sliced and preprocessed on the host (its column span comes back with its key), once.
//
// Applicable when ptr is non-decreasing (disjoint node ranges) and every graph has at most UGS_BATCH_PASS_MAX_N vertices and
// UGS_BATCH_PASS_MAX_COLS columns -- up to there the key covers a graph's whole content, so a cached graph with the same key
// has exactly the CSR the pass has just built (beyond, the reference's key samples the columns and the general path below
// reproduces what it then does: it samples from the CACHED graph).  Everything else takes the general path.
// UGS_DEVICE_BATCH=0 switches the pass off, =1 forces it wherever it applies (see device_batch_mode).
// ---------------------------------------------------------------------------------------------------------------
struct RootArena {
    UgsRootRec *roots = nullptr; int64_t roots_cap = 0, roots_used = 0;
    int2 *via = nullptr; int64_t via_cap = 0, via_used = 0;
    std::map<int64_t, std::vector<int64_t>> free_roots, free_via;      // exact-size reuse of released entries
    void *pinned = nullptr; size_t pinned_bytes = 0;                    // staging: the batch going up, the keys coming back, two descriptor slots
    hipEvent_t desc_ev[2] = {nullptr, nullptr}; int desc_slot = 0;
    unsigned long long *d_bump = nullptr;                               // device counter handing out CSR space to the graphs of a pass (never reset)
    unsigned long long bump_host = 0;                                   // its value before the next pass
    unsigned long long done_host = 0;                                   // likewise the blocks-done counter (d_bump + 1) behind the kernels' completion signal
    uint32_t epoch = 0;                                                 // names a pass: written to the flag word by a graph beyond the limits
    std::mutex call_mu;                                                 // one device pass at a time per device (shared staging)
};
std::mutex g_arena_mu;
std::map<int, RootArena *> &g_arenas = *new std::map<int, RootArena *>();
constexpr int64_t kArenaRoots = (int64_t)2 << 20, kArenaVia = (int64_t)1 << 20;     // 48 MB of root records + 8 MB of viable entries per device

RootArena *arena_of(int dev) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    auto it = g_arenas.find(dev);
    if (it != g_arenas.end()) return it->second;
    auto *a = new RootArena();
    if (hipMalloc(reinterpret_cast<void **>(&a->roots), (size_t)kArenaRoots * sizeof(UgsRootRec)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&a->via), (size_t)kArenaVia * sizeof(int2)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&a->d_bump), 256) != hipSuccess || hipMemset(a->d_bump, 0, 256) != hipSuccess) {
        (void)hipGetLastError();
        if (a->roots) (void)hipFree(a->roots);
        delete a;
        g_arenas[dev] = nullptr;                                        // no arena on this device: the general path serves every call
        return nullptr;
    }
    a->roots_cap = kArenaRoots; a->via_cap = kArenaVia;
    g_arenas[dev] = a;
    return a;
}
int64_t arena_take(std::map<int64_t, std::vector<int64_t>> &fl, int64_t &used, int64_t cap, int64_t n) {   // g_arena_mu held
    if (n <= 0) return 0;
    auto it = fl.find(n);
    if (it != fl.end() && !it->second.empty()) { const int64_t off = it->second.back(); it->second.pop_back(); return off; }
    if (used + n > cap) return -1;
    const int64_t off = used;
    used += n;
    return off;
}
void arena_release(int dev, int64_t roots_off, int64_t n_roots, int64_t via_off, int64_t n_via) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    auto it = g_arenas.find(dev);
    if (it == g_arenas.end() || !it->second) return;
    if (n_roots > 0 && roots_off >= 0) it->second->free_roots[n_roots].push_back(roots_off);
    if (n_via > 0 && via_off >= 0) it->second->free_via[n_via].push_back(via_off);
}
// the graph's root records (level 0) or viable list (levels 1, 2) in the device's arena; uploaded once per graph and device
int graph_dev_roots(Graph &g, int dev, RootArena *ar, hipStream_t s, Graph::DevRoots &out) {
    std::lock_guard<std::mutex> lk(g.pre_mu);
    for (auto &d : g.dev_roots) if (d.dev == dev) { out = d; return UGS_OK; }
    if (!g.host_ready) return UGS_E_UNSUPPORTED;                        // a stub of another device's pass: the general path completes it first
    Graph::DevRoots d{dev, -1, 0, -1, 0};
    if (g.level == 0) d.n_roots = g.n; else d.n_via = (int64_t)g.viable.size();
    {
        std::lock_guard<std::mutex> lk2(g_arena_mu);
        d.roots_off = arena_take(ar->free_roots, ar->roots_used, ar->roots_cap, d.n_roots);
        d.via_off = arena_take(ar->free_via, ar->via_used, ar->via_cap, d.n_via);
    }
    if (d.roots_off < 0 || d.via_off < 0) {
        if (d.roots_off >= 0 || d.via_off >= 0) arena_release(dev, d.roots_off, d.n_roots, d.via_off, d.n_via);
        return UGS_E_UNSUPPORTED;                                       // arena full: not an error, the caller takes the general path
    }
    if (g.level == 0 && g.n > 0) {
        std::vector<UgsRootRec> rec((size_t)g.n);
        for (int64_t vi = 0; vi < g.n; ++vi) {
            UgsRootRec &r = rec[(size_t)vi];
            r.prob = g.prob[(size_t)vi]; r.alias = g.alias[(size_t)vi];
            r.v_self = g.order[(size_t)vi]; r.v_alias = g.order[(size_t)g.alias[(size_t)vi]]; r.pad = 0;
        }
        HIP_TRY(hipMemcpyAsync(ar->roots + d.roots_off, rec.data(), rec.size() * sizeof(UgsRootRec), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));                               // `rec` is pageable and goes out of scope
    } else if (g.level > 0 && !g.viable.empty()) {
        std::vector<int2> v(g.viable.size());
        for (size_t t = 0; t < g.viable.size(); ++t) v[t] = make_int2(g.viable[t], g.order[(size_t)g.viable[t]]);
        HIP_TRY(hipMemcpyAsync(ar->via + d.via_off, v.data(), v.size() * sizeof(int2), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    g.dev_roots.push_back(d);
    out = d;
    return UGS_OK;
}

// UGS_DEVICE_BATCH: 0 = never, 1 = whenever applicable, unset = whenever applicable and the batch has at least
// kBatchPassMinCols columns.  The chain upload -> kernel -> keys back costs ~20 us of latencies whatever the size (round 4: the host
// polls the kernel's completion word instead of waiting on the stream, -5 us).  Measured with the two paths taking turns call by call
// (tools/batch_pass_probe.py, medians of 55 calls each, host-visible / device outputs): 4672 columns (PROTEINS-shaped) 0.261 / 0.175
// against 0.279 / 0.187 ms for the host's own pass + plan assembly: the pass wins; ~1200 columns (QM9- / MUTAG-shaped batches of 32
// graphs) 0.772 / 0.193 against 0.742 / 0.177 ms and 0.134 / 0.135 against 0.118 / 0.119 ms: it still loses ~16 us.  (Sequential
// measurements -- first one path, then the other -- had shown a tie there: the order of the two had favoured the second.)
constexpr int64_t kBatchPassMinCols = 2048;
int device_batch_mode() {
    const char *e = std::getenv("UGS_DEVICE_BATCH");
    if (e && e[0] == '0') return 0;
    if (e && e[0] == '1') return 1;
    return 2;
}

constexpr int kBatchNotApplicable = 1;      // positive: not an error code of the C ABI

// 64-byte slots of pinned host memory, handed out from slabs that live as long as the process
std::mutex g_pin_mu;
std::vector<char *> &g_pin_free = *new std::vector<char *>();
char *pin_slot_get() {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_free.empty()) {
        void *slab = nullptr;
        if (hipHostMalloc(&slab, 64 * 256, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        std::memset(slab, 0, 64 * 256);
        for (int i = 255; i >= 0; --i) g_pin_free.push_back(static_cast<char *>(slab) + 64 * i);
    }
    char *p = g_pin_free.back();
    g_pin_free.pop_back();
    std::memset(p, 0, 64);                      // whatever the slot's last owner's kernels wrote (they are long over)
    return p;
}
std::atomic<uint32_t> g_pin_epoch{0};           // one sequence for all plans: a slot changes hands, its signal values never repeat
void pin_slot_put(char *p) { if (p) { std::lock_guard<std::mutex> lk(g_pin_mu); g_pin_free.push_back(p); } }

// Wait for a kernel that signals its end by writing `want` to a word in pinned host memory (ugs_batch.hip: bp_signal_done): polling the
// word returns ~5 us earlier than hipStreamSynchronize on this stack (tools/sync_probe.hip: 5.9 against 10.9 us for an empty kernel).
// The budget bounds the spin: a kernel that faulted never signals, and the stream wait behind the spin reports its error.
hipError_t wait_signal(const volatile uint32_t *word, uint32_t want, hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0;; ++it) {
        if (*word == want) { std::atomic_thread_fence(std::memory_order_acquire); return hipSuccess; }
        if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
    }
    return hipStreamSynchronize(s);
}
std::atomic<int64_t> g_bp_plans{0}, g_bp_fallbacks{0};     // plans built by the device pass / calls it handed to the general path

// Returns UGS_OK with *plan_out set, kBatchNotApplicable (the caller takes the general path; the LRU has not been touched, or
// its counters have been put back), or an error.
int device_batch_plan(const int64_t *src, const int64_t *dst, int64_t E, const int64_t *ptr, int64_t G, int k, const DeviceCtx &dc,
                      bool have_hash, const Hash128 &bh, ugs_plan **plan_out, std::vector<std::pair<uint64_t, int64_t>> &touched) {
    if (G <= 0 || E <= 0 || G > ((int64_t)1 << 20) || E >= ((int64_t)1 << 24)) return kBatchNotApplicable;
    RootArena *ar = arena_of(dc.id);
    if (!ar) return kBatchNotApplicable;
    std::lock_guard<std::mutex> call_lk(ar->call_mu);
    // pinned staging, one buffer: [src E | dst E | ptr G+1 | rstart G] goes up in ONE copy, [keys G | cnt G | jminc G | jmax G | flag] comes
    // back in one, two descriptor slots go up behind the LRU replay
    const size_t up_words = (size_t)(2 * E + 2 * G + 1), back_bytes = (size_t)G * 20 + 16, desc_bytes = (size_t)G * sizeof(UgsGraphDesc);
    const size_t miss_bytes = (size_t)G * (sizeof(UgsBpMissIn) + sizeof(UgsBpMissOut)) + 16;
    const size_t st_back = align_up(up_words * 8), st_desc = align_up(st_back + back_bytes), st_miss = align_up(st_desc + 2 * align_up(desc_bytes)),
                 st_total = align_up(st_miss + miss_bytes);
    // the staging's layout depends on this call's E and G: a descriptor copy of an earlier call (asynchronous, from a region placed by
    // ITS sizes) must be over before anything here is written.  Normally long done.
    for (auto &ev : ar->desc_ev) if (ev) (void)hipEventSynchronize(ev);
    if (ar->pinned_bytes < st_total) {
        if (ar->pinned) { for (auto &ev : ar->desc_ev) if (ev) (void)hipEventSynchronize(ev); (void)hipHostFree(ar->pinned); }
        ar->pinned = nullptr; ar->pinned_bytes = 0;
        const size_t want = std::max<size_t>(st_total * 2, 1 << 18);
        if (hipHostMalloc(&ar->pinned, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return kBatchNotApplicable; }
        ar->pinned_bytes = want;
        std::memset(ar->pinned, 0, want);
    }
    auto *hostv = static_cast<int64_t *>(ar->pinned);                   // src | dst | ptr | rstart
    int64_t *h_ptr = hostv + 2 * E, *h_rstart = h_ptr + (G + 1);
    int64_t rows_total = 0, nv = 0;
    for (int64_t g = 0; g < G; ++g) {
        const int64_t n = ptr[g + 1] - ptr[g];
        if (n < 0 || n > UGS_BATCH_PASS_MAX_N) return kBatchNotApplicable;      // non-monotone ptr, or a large graph
        h_rstart[g] = rows_total;
        if (n > 0 && n >= k) { rows_total += n + 1; nv += n; }
    }
    std::memcpy(hostv, src, (size_t)E * 8);
    std::memcpy(hostv + E, dst, (size_t)E * 8);
    std::memcpy(h_ptr, ptr, (size_t)(G + 1) * sizeof(int64_t));
    // ---- one device allocation: the plan's arrays, then the pass's inputs and scratch -------------------------------------------
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 8)); return o; };
    const size_t o_desc = take(desc_bytes), o_row = take((size_t)rows_total * sizeof(int64_t));
    const size_t o_adj = take((size_t)2 * E * sizeof(int2)), o_adjf = take((size_t)2 * E * sizeof(int2));
    const bool fused = G * E <= ugs_batch_pass_fused_work();
    const size_t o_vrank = take((size_t)rows_total * sizeof(int32_t));
    const size_t o_up = take(up_words * 8), o_owner = take(fused ? 8 : (size_t)E * 4), o_cnt = take(fused ? 8 : (size_t)G * 12);
    auto *p = new ugs_plan();
    if (int rc = pool_get(off, dc.id, p->blob_buf)) { delete p; return rc; }
    char *base = static_cast<char *>(p->blob_buf.p);
    auto bail = [&](int rc) { pool_put(p->blob_buf); delete p; return rc; };
    hipStream_t s = dc.stream;
    Lap lap;
    auto *d_src = reinterpret_cast<int64_t *>(base + o_up), *d_dst = d_src + E, *d_ptr = d_dst + E;
    // what comes back is written by the kernel straight into the pinned staging (keys[G] u64 | cnt[G] | jminc[G] | jmax[G] | flag): no
    // copy command behind the kernel, the host waits for the stream only
    char *h_back = static_cast<char *>(ar->pinned) + st_back;
    const uint32_t epoch = ++ar->epoch ? ar->epoch : ++ar->epoch;       // never 0
    *reinterpret_cast<volatile uint32_t *>(h_back + (size_t)G * 20) = 0u;
    *reinterpret_cast<volatile uint32_t *>(h_back + (size_t)G * 20 + 4) = 0u;     // the completion word: the staging's layout moves with E and G, so the
                                                                                   // word may hold anything an earlier call left there -- e.g. a column count equal to this epoch
    hipError_t e = hipMemcpyAsync(d_src, hostv, up_words * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = ugs_launch_batch_pass(d_src, d_dst, E, d_ptr, G, k, reinterpret_cast<int32_t *>(base + o_owner), reinterpret_cast<uint32_t *>(base + o_cnt),
                                                   d_ptr + (G + 1), reinterpret_cast<int64_t *>(base + o_row), reinterpret_cast<int2 *>(base + o_adj),
                                                   reinterpret_cast<int2 *>(base + o_adjf), reinterpret_cast<int32_t *>(base + o_vrank), ar->d_bump, ar->bump_host,
                                                   epoch, h_back, ar->d_bump + 1, ar->done_host, s);
    if (e == hipSuccess) e = wait_signal(reinterpret_cast<volatile uint32_t *>(h_back + (size_t)G * 20 + 4), epoch, s);
    if (e != hipSuccess) {       // the counters' values are unknown now: start again from zero
        (void)hipDeviceSynchronize();
        (void)hipMemset(ar->d_bump, 0, 16); ar->bump_host = 0; ar->done_host = 0;
        return bail(fail_hip(e, "device batch pass"));
    }
    ar->done_host += (unsigned long long)G;
    const double t_pass = lap();
    const auto *h_keys = reinterpret_cast<const unsigned long long *>(h_back);
    const auto *h_cnt = reinterpret_cast<const uint32_t *>(h_back + (size_t)G * 8), *h_jminc = h_cnt + G, *h_jmax = h_jminc + G;
    {   // the counter moved by twice the columns of the graphs that got as far as their allocation: re-derive it from what came back
        unsigned long long used = 0;
        for (int64_t g = 0; g < G; ++g) used += 2ull * h_cnt[g];
        if (h_jmax[G] == epoch) {                                        // a graph beyond the limits of the pass: which blocks allocated is not known
            (void)hipStreamSynchronize(s);                              // the signal came from the last block's thread 0: let the kernel retire before its counter is rewritten
            (void)hipMemset(ar->d_bump, 0, 8); ar->bump_host = 0;
            return bail(kBatchNotApplicable);
        }
        ar->bump_host += used;
    }
    // ---- the LRU replay on G keys: the same lookups, in graph order, as the general path ----------------------------------------
    const int slot = ar->desc_slot ^= 1;                                 // the copy of the call before last on this device is long over; wait if not
    if (!ar->desc_ev[slot]) { if (hipEventCreateWithFlags(&ar->desc_ev[slot], hipEventDisableTiming) != hipSuccess) return bail(fail(UGS_E_HIP, "hipEventCreate")); }
    else (void)hipEventSynchronize(ar->desc_ev[slot]);
    UgsGraphDesc *desc = reinterpret_cast<UgsGraphDesc *>(static_cast<char *>(ar->pinned) + st_desc + (size_t)slot * align_up(desc_bytes));
    std::vector<int64_t> evicted, ru, rv;
    std::vector<std::shared_ptr<Graph>> graphs((size_t)G);
    int64_t hits = 0, misses = 0, owned_cols = 0;
    p->g_n.assign((size_t)G, 0); p->g_maxdeg.assign((size_t)G, 0); p->g_sbdeg.assign((size_t)G, 0.0); p->g_level.assign((size_t)G, -1);
    touched.clear();
    // Graphs the LRU does not know (cold path).  Up to UGS_BATCH_ROOTS_MAX_N vertices the rest of their preprocessing runs on the
    // device as well (ugs_bp_roots, one launch for all of them behind this loop): the graph exists on the host as a stub.
    // UGS_DEVICE_COLD=0 keeps the host preprocessing (A/B, tests of both).
    struct Pending { int64_t g; std::shared_ptr<Graph> gr; uint64_t key; int64_t handle; int64_t roots_off, via_off; };
    std::vector<Pending> pend;
    std::vector<int32_t> pend_of((size_t)G, -1);
    std::vector<std::pair<uint64_t, int64_t>> inserted;                  // (key, handle) this call has put into the LRU
    const bool cold_on_device = [] { const char *e = std::getenv("UGS_DEVICE_COLD"); return !(e && e[0] == '0'); }();
    auto give_up = [&] {                                                // put the LRU back as it was: the general path repeats the lookups (same final recency order)
        {
            std::lock_guard<std::mutex> lk(g_lru_mu);
            lru().hits -= hits; lru().misses -= misses;
            if (evicted.empty()) for (auto &kv : inserted) lru().erase(kv.first);      // (an eviction cannot be undone: then the entries stay, as before)
        }
        if (evicted.empty()) for (auto &kv : inserted) drop(kv.second);
        for (auto &pd : pend) arena_release(dc.id, pd.roots_off, pd.gr->n, pd.via_off, pd.gr->n);
        for (int64_t h : evicted) drop(h);
        return bail(kBatchNotApplicable);
    };
    for (int64_t g = 0; g < G; ++g) {
        UgsGraphDesc &d = desc[(size_t)g];
        const int64_t lo = ptr[g], n = ptr[g + 1] - lo;
        d.node_lo = lo; d.rbase = h_rstart[g]; d.vbase = 0; d.viable_base = 0; d.n = 0; d.level = -1; d.n_viable = 0; d.pad = 0;
        if (n <= 0 || n < k) continue;                                   // degenerate: m rows of -1 (never looked up: reference :132-143)
        owned_cols += h_cnt[g];
        const uint64_t key = (uint64_t)h_keys[g];
        int64_t handle = 0;
        bool hit;
        {
            std::lock_guard<std::mutex> lk(g_lru_mu);
            hit = lru().get(key, handle);
            if (hit) ++lru().hits; else ++lru().misses;
        }
        std::shared_ptr<Graph> gr = hit ? lookup(handle) : nullptr;
        if (hit) ++hits; else ++misses;
        if (!gr && cold_on_device && n <= UGS_BATCH_ROOTS_MAX_N) {       // unknown graph: a stub now, its root records by the launch behind this loop
            int64_t ro, vo;
            {
                std::lock_guard<std::mutex> lk2(g_arena_mu);
                ro = arena_take(ar->free_roots, ar->roots_used, ar->roots_cap, n);
                vo = arena_take(ar->free_via, ar->via_used, ar->via_cap, n);
            }
            if (ro < 0 || vo < 0) {
                arena_release(dc.id, ro, ro >= 0 ? n : 0, vo, vo >= 0 ? n : 0);
                return give_up();                                        // arena full: the general path serves this batch
            }
            gr = std::make_shared<Graph>();
            gr->n = n; gr->nnz = 2 * (int64_t)h_cnt[g]; gr->k_built = k; gr->host_ready = false; gr->stub_lo = lo;
            if (h_cnt[g]) {                                              // its span of the columns, kept until some host path needs the arrays
                const int64_t j0 = (int64_t)(0xFFFFFFFFu - h_jminc[g]), j1 = (int64_t)h_jmax[g] + 1;
                gr->stub_cols.resize((size_t)(2 * (j1 - j0)));
                std::memcpy(gr->stub_cols.data(), src + j0, (size_t)(j1 - j0) * 8);
                std::memcpy(gr->stub_cols.data() + (j1 - j0), dst + j0, (size_t)(j1 - j0) * 8);
            }
            handle = enroll(gr);
            {
                int64_t ev = 0;
                std::lock_guard<std::mutex> lk(g_lru_mu);
                if (lru().put(key, handle, ev)) evicted.push_back(ev);
            }
            inserted.emplace_back(key, handle);
            pend_of[(size_t)g] = (int32_t)pend.size();
            pend.push_back(Pending{g, gr, key, handle, ro, vo});
            touched.emplace_back(key, handle);
            graphs[(size_t)g] = gr;
            continue;
        }
        if (gr && !gr->host_ready) {                                     // a stub: of this very call (a repeated graph), or of an earlier one
            bool mine = false;
            for (size_t i = 0; i < pend.size() && !mine; ++i) if (pend[i].gr.get() == gr.get()) { pend_of[(size_t)g] = (int32_t)i; mine = true; }
            if (mine) {
                if (gr->n != n || gr->nnz != 2 * (int64_t)h_cnt[g]) return give_up();
                touched.emplace_back(key, handle);
                graphs[(size_t)g] = gr;
                continue;
            }
        }
        if (!gr) {                                                       // unknown graph: sliced and preprocessed on the host, once
            ru.clear(); rv.clear();
            if (h_cnt[g]) for (int64_t j = (int64_t)(0xFFFFFFFFu - h_jminc[g]); j <= (int64_t)h_jmax[g]; ++j) {
                const int64_t u = src[j], v = dst[j];
                if (u >= lo && u < lo + n && v >= lo && v < lo + n) { ru.push_back(u - lo); rv.push_back(v - lo); }
            }
            if ((int64_t)ru.size() != (int64_t)h_cnt[g]) return give_up();
            if (int rc = make_graph(ru.data(), rv.data(), (int64_t)ru.size(), n, k, gr)) { give_up(); return rc; }
            handle = enroll(gr);
            int64_t ev = 0;
            std::lock_guard<std::mutex> lk(g_lru_mu);
            if (lru().put(key, handle, ev)) evicted.push_back(ev);
            inserted.emplace_back(key, handle);
        }
        if (gr->n != n || gr->nnz != 2 * (int64_t)h_cnt[g]) return give_up();   // a 64-bit key collision: let the general path do what the reference does
        Graph::DevRoots dr{};
        if (int rc = graph_dev_roots(*gr, dc.id, ar, s, dr)) { if (rc == UGS_E_UNSUPPORTED) return give_up(); give_up(); return rc; }
        d.n = (int32_t)n; d.level = gr->level; d.n_viable = (int32_t)gr->n_viable;
        d.vbase = gr->level == 0 ? dr.roots_off : 0; d.viable_base = gr->level > 0 ? dr.via_off : 0;
        p->g_n[(size_t)g] = n; p->g_maxdeg[(size_t)g] = gr->max_deg; p->g_sbdeg[(size_t)g] = gr->sb_deg; p->g_level[(size_t)g] = gr->level;
        touched.emplace_back(key, handle);
        graphs[(size_t)g] = gr;
    }
    if (!pend.empty()) {                                                 // ---- the unknown graphs: root records on the device, 32 bytes each back
        auto *h_in = reinterpret_cast<UgsBpMissIn *>(static_cast<char *>(ar->pinned) + st_miss);
        auto *h_out = reinterpret_cast<UgsBpMissOut *>(h_in + G);
        for (size_t i = 0; i < pend.size(); ++i) h_in[i] = UgsBpMissIn{(int32_t)pend[i].g, 0, pend[i].roots_off, pend[i].via_off};
        auto *h_rdone = reinterpret_cast<uint32_t *>(h_out + G);
        *reinterpret_cast<volatile uint32_t *>(h_rdone) = 0u;
        e = ugs_launch_batch_roots(d_ptr, d_ptr + (G + 1), reinterpret_cast<const int64_t *>(base + o_row), reinterpret_cast<const int2 *>(base + o_adj),
                                   reinterpret_cast<const int32_t *>(base + o_vrank), h_in, h_out, (int64_t)pend.size(), k, ar->roots, ar->via,
                                   ar->d_bump + 1, ar->done_host, h_rdone, epoch, s);
        if (e == hipSuccess) e = wait_signal(h_rdone, epoch, s);
        if (e != hipSuccess) {
            (void)hipDeviceSynchronize();
            (void)hipMemset(ar->d_bump, 0, 16); ar->bump_host = 0; ar->done_host = 0;
            give_up();
            return fail_hip(e, "device batch pass: root records");
        }
        ar->done_host += (unsigned long long)pend.size();
        for (size_t i = 0; i < pend.size(); ++i) {
            Graph &gg = *pend[i].gr;
            const UgsBpMissOut &o = h_out[i];
            const int64_t n = gg.n;
            std::lock_guard<std::mutex> lk(gg.pre_mu);
            if (!gg.host_ready) {                                        // (a host path that met the stub meanwhile has filled in the same values)
                gg.level = o.level; gg.n_viable = o.n_viable; gg.nonzero = o.nonzero; gg.Z = o.Z; gg.max_deg = o.max_deg; gg.sb_deg = o.sb_deg;
            }
            if (o.level == 0) { gg.dev_roots.push_back(Graph::DevRoots{dc.id, pend[i].roots_off, n, -1, 0}); arena_release(dc.id, -1, 0, pend[i].via_off, n); }
            else { gg.dev_roots.push_back(Graph::DevRoots{dc.id, -1, 0, pend[i].via_off, n}); arena_release(dc.id, pend[i].roots_off, n, -1, 0); }
        }
        for (int64_t g = 0; g < G; ++g) {
            if (pend_of[(size_t)g] < 0) continue;
            const Pending &pd = pend[(size_t)pend_of[(size_t)g]];
            const Graph &gg = *pd.gr;
            UgsGraphDesc &d = desc[(size_t)g];
            d.n = (int32_t)gg.n; d.level = gg.level; d.n_viable = (int32_t)gg.n_viable;
            d.vbase = gg.level == 0 ? pd.roots_off : 0; d.viable_base = gg.level > 0 ? pd.via_off : 0;
            p->g_n[(size_t)g] = gg.n; p->g_maxdeg[(size_t)g] = gg.max_deg; p->g_sbdeg[(size_t)g] = gg.sb_deg; p->g_level[(size_t)g] = gg.level;
        }
        pend.clear();
    }
    for (int64_t h : evicted) drop(h);                                   // only evicted handles are destroyed; cached ones live on
    evicted.clear();
    if (debug_on()) {
        std::lock_guard<std::mutex> lk(g_lru_mu);
        std::fprintf(stderr, "[UGS CACHE] hits=%lld misses=%lld cache_size=%zu\n", (long long)hits, (long long)misses, lru().items.size());
    }
    const double t_lru = lap();
    e = hipMemcpyAsync(base + o_desc, desc, desc_bytes, hipMemcpyHostToDevice, s);      // pinned source: truly asynchronous; the plan's first
    if (e == hipSuccess) e = hipEventRecord(ar->desc_ev[slot], s);                        // walk is ordered behind it on the plan (plan_enter)
    if (e != hipSuccess) return bail(fail_hip(e, "plan descriptors"));
    if (int rc = plan_leave(p, s)) return bail(rc);                      // a first call on another stream waits for the descriptors (plan_enter)
    if (std::getenv("UGS_BP_TRACE"))
        std::fprintf(stderr, "[UGS BATCH PASS] G=%lld E=%lld: upload + pass + keys back %.1f us, LRU replay + arena %.1f us, descriptors %.1f us\n",
                     (long long)G, (long long)E, t_pass * 1e6, t_lru * 1e6, lap() * 1e6);
    for (auto &gr : graphs) if (gr) p->keep.push_back(gr);
    p->device = dc.id; p->cus = dc.cus;
    p->G = G; p->nverts = nv; p->nnz = 2 * owned_cols;
    p->blob = base; p->blob_bytes = off;
    p->dev.graphs = reinterpret_cast<const UgsGraphDesc *>(base + o_desc);
    p->dev.rowptr = reinterpret_cast<const int64_t *>(base + o_row);
    p->dev.adj = reinterpret_cast<const int2 *>(base + o_adj);
    p->dev.adjf = reinterpret_cast<const int2 *>(base + o_adjf);
    p->dev.roots = ar->roots;
    p->dev.viable = ar->via;
    p->dev.num_graphs = G;
    p->prow_failed = true;                       // vbase indexes the arena, not a per-plan vertex range: these plans read rows through rowptr
    if (have_hash) {
        p->cache_key = mum(bh.a ^ 0x6a09e667f3bcc909ull, bh.b ^ (uint64_t)k) ^ 0xd6e8feb86659fd93ull;
        plan_cache_put(p);
    }
    *plan_out = p;
    return UGS_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" {

const char *ugs_last_error(void) { return t_err.c_str(); }
const char *ugs_version(void) { return "ugs-mi355 0.1 (gfx950)"; }

int ugs_device_count(int *count) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) { if (count) *count = 0; return fail(UGS_E_NO_DEVICE, std::string("no usable HIP device: ") + hipGetErrorString(e)); }
    if (count) *count = c;
    return UGS_OK;
}

int ugs_set_device(int device) {
    int c = 0;
    if (int rc = ugs_device_count(&c)) return rc;
    if (device < 0 || device >= c) return fail(UGS_E_BAD_ARG, "device index out of range");
    t_device = device;
    HIP_TRY(hipSetDevice(device));
    return UGS_OK;
}

int ugs_set_stream(void *stream, int use) {
    t_job_stream = static_cast<hipStream_t>(stream);
    t_job_stream_set = use != 0;
    return UGS_OK;
}

int ugs_create_preproc(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, int64_t num_nodes, int k, int64_t *handle_out) {
    if (!handle_out || (num_cols > 0 && !edge_index) || num_cols < 0) return fail(UGS_E_BAD_ARG, "bad arguments to create_preproc");
    std::shared_ptr<Graph> g;
    if (int rc = make_graph(edge_index, edge_index + row_stride, num_cols, num_nodes, k, g)) return rc;
    *handle_out = enroll(std::move(g));
    return UGS_OK;
}

int ugs_destroy_preproc(int64_t handle) { drop(handle); return UGS_OK; }

int ugs_has_graphlets(int64_t handle, int *out) {
    auto g = lookup(handle);
    if (out) *out = (g && g->Z > 0.0) ? 1 : 0;
    return UGS_OK;
}

int ugs_get_preproc_info(int64_t handle, int *found, int64_t *num_nodes, int64_t *num_edges_stored, double *Z, int *bucket_count_nonzero) {
    auto g = lookup(handle);
    if (found) *found = g ? 1 : 0;
    if (!g) return UGS_OK;
    if (num_nodes) *num_nodes = g->n;
    if (num_edges_stored) *num_edges_stored = g->nnz;
    if (Z) *Z = g->Z;
    if (bucket_count_nonzero) *bucket_count_nonzero = g->nonzero;
    return UGS_OK;
}

int ugs_preproc_dump(int64_t handle, int64_t *indptr, int32_t *indices, int32_t *edge_col, int32_t *order, int32_t *index_of,
                     int32_t *suffix_deg, double *bucket_b, double *prob, int32_t *alias) {
    auto g = lookup(handle);
    if (!g) return fail(UGS_E_INVALID_HANDLE, "Invalid preproc handle");
    if (!g->host_ready) complete_on_host(*g);
    const size_t n = (size_t)g->n, nnz = (size_t)g->nnz;
    if (indptr) std::memcpy(indptr, g->rowptr.data(), (n + 1) * sizeof(int64_t));
    if (indices && nnz) std::memcpy(indices, g->nbr.data(), nnz * sizeof(int32_t));
    if (edge_col && nnz) std::memcpy(edge_col, g->col.data(), nnz * sizeof(int32_t));
    if (order && n) std::memcpy(order, g->order.data(), n * sizeof(int32_t));
    if (index_of && n) std::memcpy(index_of, g->rank.data(), n * sizeof(int32_t));
    if (suffix_deg && n) std::memcpy(suffix_deg, g->sdeg.data(), n * sizeof(int32_t));
    if (bucket_b && n) std::memcpy(bucket_b, g->weight.data(), n * sizeof(double));
    if (prob && n) std::memcpy(prob, g->prob.data(), n * sizeof(double));
    if (alias && n) std::memcpy(alias, g->alias.data(), n * sizeof(int32_t));
    return UGS_OK;
}

int ugs_cache_clear(void) {
    {
        std::lock_guard<std::mutex> lk(g_lru_mu);
        Lru &c = lru();
        for (auto &kv : c.items) drop(kv.second);
        c.items.clear(); c.index.clear(); c.hits = c.misses = 0;
    }
    plan_cache_clear();
    batch_index_clear();
    return UGS_OK;
}

int ugs_batch_pass_stats(int64_t *device_plans, int64_t *general_path) {
    if (device_plans) *device_plans = g_bp_plans.load();
    if (general_path) *general_path = g_bp_fallbacks.load();
    return UGS_OK;
}

int ugs_cache_stats(int64_t *size, int64_t *hits, int64_t *misses) {
    std::lock_guard<std::mutex> lk(g_lru_mu);
    Lru &c = lru();
    if (size) *size = (int64_t)c.items.size();
    if (hits) *hits = c.hits;
    if (misses) *misses = c.misses;
    return UGS_OK;
}

// ---- plans ------------------------------------------------------------------------------------------------------
int ugs_plan_create_handle(int64_t handle, ugs_plan **plan_out) {
    if (!plan_out) return fail(UGS_E_BAD_ARG, "plan_out is null");
    auto g = lookup(handle);
    if (!g) return fail(UGS_E_INVALID_HANDLE, "Invalid preproc handle");
    if (!g->host_ready) complete_on_host(*g);
    DeviceCtx dc;
    if (int rc = device_ctx(dc)) return rc;
    std::lock_guard<std::mutex> lk(g->plan_mu);
    if (g->plan && g->plan->device == dc.id) { g->plan->refs.fetch_add(1); *plan_out = g->plan; return UGS_OK; }
    auto *p = new ugs_plan();
    p->handle_api = true;
    std::vector<PlanPiece> pieces(1);
    pieces[0].g = g;                       // note: the plan keeps the Graph alive only through its arrays on the device
    if (int rc = assemble_plan(pieces, dc, p)) { delete p; return rc; }
    if (!g->plan) { g->plan = p; p->refs.fetch_add(1); }
    *plan_out = p;
    return UGS_OK;
}

int ugs_plan_create_batch(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr, int64_t num_graphs,
                          int k, ugs_plan **plan_out) {
    if (!plan_out || !ptr || num_graphs < 0 || num_cols < 0 || (num_cols > 0 && !edge_index)) return fail(UGS_E_BAD_ARG, "bad arguments to sample_batch");
    if (num_cols >= (int64_t)INT32_MAX) return fail(UGS_E_UNSUPPORTED, "batch too large: columns must be < 2^31 - 1");
    DeviceCtx dc;
    if (int rc = device_ctx(dc)) return rc;
    const int64_t G = num_graphs, E = num_cols;
    const int64_t *src = edge_index, *dst = edge_index + row_stride;
    // the batch as a whole: seen before, and everything it needs still cached?
    const bool use_index = std::getenv("UGS_NO_BATCH_INDEX") == nullptr;
    Hash128 bh{0, 0};
    Witness wit{};
    if (use_index) {
        bh = batch_hash(src, dst, E, ptr, G);
        wit = batch_witness(src, dst, E, ptr, G);
        BatchEntry found;
        bool have = false;
        {
            std::lock_guard<std::mutex> lk(g_bi_mu);
            for (auto it = g_batch_index.begin(); it != g_batch_index.end(); ++it)
                if (it->h == bh && it->E == E && it->G == G && it->k == k && it->dev == dc.id && it->wit == wit) {
                    g_batch_index.splice(g_batch_index.begin(), g_batch_index, it);
                    found = *it; have = true;
                    break;
                }
        }
        if (have) {
            int64_t hits = 0;
            if (batch_replay(found, hits)) {
                if (ugs_plan *cached = plan_cache_get(found.plan_key, dc.id)) {
                    if (debug_on()) {
                        std::lock_guard<std::mutex> lk(g_lru_mu);
                        std::fprintf(stderr, "[UGS CACHE] hits=%lld misses=0 cache_size=%zu\n", (long long)hits, lru().items.size());
                    }
                    *plan_out = cached;
                    return UGS_OK;
                }
            }
            // fall through: the general path repeats the lookups (same final recency order); undo the hits counted above
            std::lock_guard<std::mutex> lk(g_lru_mu);
            lru().hits -= hits;
        }
    }
    const int bp_mode = device_batch_mode();
    // A workload the pass cannot take (a graph beyond its limits in every batch, a full arena) would pay upload + pass + wait before
    // every general-path call: after three refusals in a row the default mode leaves the pass alone for the next 64 calls.
    static std::atomic<int> bp_refused{0}, bp_pause{0};
    const bool bp_paused = bp_mode == 2 && bp_pause.load() > 0 && bp_pause.fetch_sub(1) > 0;
    if (!bp_paused && (bp_mode == 1 || (bp_mode == 2 && E >= kBatchPassMinCols))) {   // batches of small graphs: slicing, keys and CSR on the device
        std::vector<std::pair<uint64_t, int64_t>> touched_d;
        ugs_plan *dp = nullptr;
        const int rc = device_batch_plan(src, dst, E, ptr, G, k, dc, use_index, bh, &dp, touched_d);
        if (rc == UGS_OK) {
            bp_refused.store(0);
            g_bp_plans.fetch_add(1);
            if (use_index) {
                std::lock_guard<std::mutex> lk(g_bi_mu);
                for (auto it = g_batch_index.begin(); it != g_batch_index.end(); ++it)
                    if (it->h == bh && it->E == E && it->G == G && it->k == k && it->dev == dc.id) { g_batch_index.erase(it); break; }
                g_batch_index.push_front(BatchEntry{bh, E, G, k, dc.id, dp->cache_key, std::move(touched_d), wit});
                while (g_batch_index.size() > g_plan_cache_cap) g_batch_index.pop_back();
            }
            *plan_out = dp;
            return UGS_OK;
        }
        if (rc != kBatchNotApplicable) return rc;
        g_bp_fallbacks.fetch_add(1);
        if (bp_refused.fetch_add(1) + 1 >= 3) { bp_refused.store(0); bp_pause.store(64); }
    }
    std::vector<int64_t> cstart, cols_of;                      // per-graph column lists, concatenated
    Lap lap;
    assign_columns(src, dst, E, ptr, G, cstart, cols_of);
    const double t_assign = lap();
    // --- per graph: renumber, hash, LRU lookup / preprocessing ------------------------------------------------------
    std::vector<PlanPiece> pieces((size_t)G);
    std::vector<std::pair<uint64_t, int64_t>> touched;          // (LRU key, handle) per non-degenerate graph, for the batch index
    std::vector<int64_t> ru, rv, evicted;
    uint64_t pkey = 14695981039346656037ull;
    const uint64_t prime = 1099511628211ull;
    auto mix = [&](uint64_t x) { pkey = (pkey ^ x) * prime; };
    mix((uint64_t)G);
    int64_t hits = 0, misses = 0;
    for (int64_t g = 0; g < G; ++g) {
        PlanPiece &pc = pieces[(size_t)g];
        const int64_t lo = ptr[g], hi = ptr[g + 1], n = hi - lo;
        pc.lo = lo;
        mix((uint64_t)lo);
        if (n <= 0 || n < k) { mix(0x9e3779b97f4a7c15ull); continue; }        // degenerate: m rows of -1
        const int64_t c0 = cstart[(size_t)g], cn = cstart[(size_t)g + 1] - c0;
        ru.resize((size_t)cn); rv.resize((size_t)cn);
        for (int64_t t = 0; t < cn; ++t) { const int64_t j = cols_of[(size_t)(c0 + t)]; ru[(size_t)t] = src[j] - lo; rv[(size_t)t] = dst[j] - lo; }
        const uint64_t key = graph_key(ru.data(), rv.data(), cn, n);
        int64_t handle = 0;
        bool hit;
        {
            std::lock_guard<std::mutex> lk(g_lru_mu);
            hit = lru().get(key, handle);
            if (hit) ++lru().hits; else ++lru().misses;
        }
        std::shared_ptr<Graph> gr = hit ? lookup(handle) : nullptr;
        if (!gr) {
            if (int rc = make_graph(ru.data(), rv.data(), cn, n, k, gr)) return rc;
            handle = enroll(gr);
            int64_t ev = 0;
            std::lock_guard<std::mutex> lk(g_lru_mu);
            if (lru().put(key, handle, ev)) evicted.push_back(ev);
            ++misses;
        } else {
            ++hits;
            if (!gr->host_ready) complete_on_host(*gr);
        }
        pc.g = gr;
        touched.emplace_back(key, handle);
        pc.colmap = cols_of.data() + c0;
        pc.ncols = cn;
        mix((uint64_t)handle); mix((uint64_t)cn);
        for (int64_t t = 0; t < cn; ++t) mix((uint64_t)cols_of[(size_t)(c0 + t)]);
    }
    for (int64_t h : evicted) drop(h);          // only evicted handles are destroyed; cached ones live on
    if (debug_on()) {
        std::lock_guard<std::mutex> lk(g_lru_mu);
        std::fprintf(stderr, "[UGS CACHE] hits=%lld misses=%lld cache_size=%zu\n", (long long)hits, (long long)misses, lru().items.size());
    }
    auto remember = [&] {
        if (!use_index) return;
        std::lock_guard<std::mutex> lk(g_bi_mu);
        for (auto it = g_batch_index.begin(); it != g_batch_index.end(); ++it)
            if (it->h == bh && it->E == E && it->G == G && it->k == k && it->dev == dc.id) { g_batch_index.erase(it); break; }
        g_batch_index.push_front(BatchEntry{bh, E, G, k, dc.id, pkey, std::move(touched), wit});
        while (g_batch_index.size() > g_plan_cache_cap) g_batch_index.pop_back();
    };
    if (ugs_plan *cached = plan_cache_get(pkey, dc.id)) { remember(); *plan_out = cached; return UGS_OK; }
    const double t_graphs = lap();
    auto *p = new ugs_plan();
    p->cache_key = pkey;
    if (int rc = assemble_plan(pieces, dc, p)) { delete p; return rc; }
    if (debug_on() && E >= ((int64_t)1 << 21))
        std::fprintf(stderr, "[UGS PLAN] columns -> graphs %.3fs, renumber + hash + preprocessing %.3fs, assemble + upload %.3fs\n", t_assign, t_graphs, lap());
    plan_cache_put(p);
    remember();
    *plan_out = p;
    return UGS_OK;
}

int ugs_plan_release(ugs_plan *plan) { plan_unref(plan); return UGS_OK; }

int ugs_plan_graph_roots(ugs_plan *plan, int64_t graph, int64_t capacity, int32_t *level, int32_t *num_nodes, int32_t *num_viable, double *prob,
                         int32_t *alias, int32_t *v_self, int32_t *v_alias, int32_t *viable_vi, int32_t *viable_v) {
    if (!plan || graph < 0 || graph >= plan->G) return fail(UGS_E_BAD_ARG, "graph index outside the plan");
    HIP_TRY(hipSetDevice(plan->device));
    UgsGraphDesc d{};
    {
        std::lock_guard<std::mutex> lk(plan->mu);
        if (plan->last_valid) HIP_TRY(hipEventSynchronize(plan->last_ev));
        HIP_TRY(hipMemcpy(&d, plan->dev.graphs + graph, sizeof(d), hipMemcpyDeviceToHost));
    }
    if (level) *level = d.level;
    if (num_nodes) *num_nodes = d.n;
    if (num_viable) *num_viable = d.n_viable;
    if (d.level == 0 && d.n > 0 && (prob || alias || v_self || v_alias)) {
        if (capacity < d.n) return fail(UGS_E_BAD_ARG, "capacity below the graph's vertex count");
        std::vector<UgsRootRec> rec((size_t)d.n);
        HIP_TRY(hipMemcpy(rec.data(), plan->dev.roots + d.vbase, rec.size() * sizeof(UgsRootRec), hipMemcpyDeviceToHost));
        for (int32_t vi = 0; vi < d.n; ++vi) {
            if (prob) prob[vi] = rec[(size_t)vi].prob;
            if (alias) alias[vi] = rec[(size_t)vi].alias;
            if (v_self) v_self[vi] = rec[(size_t)vi].v_self;
            if (v_alias) v_alias[vi] = rec[(size_t)vi].v_alias;
        }
    } else if (d.level > 0 && d.n_viable > 0 && (viable_vi || viable_v)) {
        if (capacity < d.n_viable) return fail(UGS_E_BAD_ARG, "capacity below the viable list's length");
        std::vector<int2> v((size_t)d.n_viable);
        HIP_TRY(hipMemcpy(v.data(), plan->dev.viable + d.viable_base, v.size() * sizeof(int2), hipMemcpyDeviceToHost));
        for (int32_t t = 0; t < d.n_viable; ++t) { if (viable_vi) viable_vi[t] = v[(size_t)t].x; if (viable_v) viable_v[t] = v[(size_t)t].y; }
    }
    return UGS_OK;
}

int ugs_plan_info(const ugs_plan *plan, int k, int64_t *num_graphs, int64_t *num_vertices, int64_t *nnz, int64_t *device_bytes, int *tier) {
    if (!plan) return fail(UGS_E_BAD_ARG, "plan is null");
    if (num_graphs) *num_graphs = plan->G;
    if (num_vertices) *num_vertices = plan->nverts;
    if (nnz) *nnz = plan->nnz;
    if (device_bytes) *device_bytes = (int64_t)(plan->blob_bytes + plan->prow.bytes);
    if (tier) *tier = choose_tier(const_cast<ugs_plan *>(plan), k).first;
    return UGS_OK;
}

static int plan_walk_impl(ugs_plan *plan, int m_per_graph, int k, int mode, int64_t extra_node_offset, int seed, const uint64_t *d_seed_ptr,
                          int64_t row_begin, int64_t row_count, void *stream, int64_t *d_nodes, int64_t *d_edge_ptr, int64_t *total_edges_host,
                          bool *defer_scan = nullptr, bool poll_total = false);


int ugs_plan_walk(ugs_plan *plan, int m_per_graph, int k, int mode, int64_t extra_node_offset, int seed, int64_t row_begin,
                  int64_t row_count, void *stream, int64_t *d_nodes, int64_t *d_edge_ptr, int64_t *total_edges_host) {
    return plan_walk_impl(plan, m_per_graph, k, mode, extra_node_offset, seed, nullptr, row_begin, row_count, stream, d_nodes, d_edge_ptr, total_edges_host);
}

// defer_scan (ugs_plan_step): where the fill kernel can turn the counts into edge_ptr itself -- first tier S (rows are filled from
// their adjacency, nothing is staged), no capture in progress -- the scan launch is left out and *defer_scan set.
static int plan_walk_impl(ugs_plan *plan, int m_per_graph, int k, int mode, int64_t extra_node_offset, int seed, const uint64_t *d_seed_ptr,
                          int64_t row_begin, int64_t row_count, void *stream, int64_t *d_nodes, int64_t *d_edge_ptr, int64_t *total_edges_host,
                          bool *defer_scan, bool poll_total) {
    if (defer_scan) *defer_scan = false;
    if (!plan) return fail(UGS_E_BAD_ARG, "plan is null");
    if (k < 1) return fail(UGS_E_BAD_ARG, "k must be >= 1");
    if (k > UGS_KMAX) return fail(UGS_E_UNSUPPORTED, "k > 32 is not supported by the HIP sampler");
    if (mode < 0 || mode > 2) return fail(UGS_E_BAD_MODE, "mode must be one of: 'sample', 'graph', 'global'");
    if (m_per_graph < 0 || row_begin < 0 || row_count < 0 || row_begin + row_count > plan->G * (int64_t)m_per_graph)
        return fail(UGS_E_BAD_ARG, "row range outside [0, num_graphs * m_per_graph)");
    if (row_count > 0 && (!d_nodes || !d_edge_ptr)) return fail(UGS_E_BAD_ARG, "null output pointer");
    HIP_TRY(hipSetDevice(plan->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (row_count == 0) {
        if (d_edge_ptr) HIP_TRY(hipMemsetAsync(d_edge_ptr, 0, sizeof(int64_t), s));
        if (total_edges_host) { HIP_TRY(hipStreamSynchronize(s)); *total_edges_host = 0; }
        return UGS_OK;
    }
    const TierChoice tc = choose_tier(plan, k);
    std::lock_guard<std::mutex> lk(plan->mu);
    if (int rc = plan_enter(plan, s)) return rc;
    if (int rc = ensure(plan->counts, (size_t)row_count * sizeof(uint32_t), plan->device, plan)) return rc;
    if (int rc = ensure(plan->scantmp, (size_t)ugs_scan_tmp_words(row_count) * sizeof(int64_t), plan->device, plan)) return rc;
    if (int rc = ensure(plan->ovfcnt, 4 * sizeof(uint32_t), plan->device, plan)) return rc;
    const bool may_overflow = tc.second >= 0 || tc.third_G;
    if (may_overflow) {
        if (int rc = ensure(plan->ovf1, (size_t)row_count * sizeof(int64_t), plan->device, plan)) return rc;
        if (int rc = ensure(plan->ovf2, (size_t)row_count * sizeof(int64_t), plan->device, plan)) return rc;
    }
    // edge staging by the walk (tiers with one walk per wave): 512 bytes of scratch per row, bounded
    const int64_t stage_max = [] { const char *e = std::getenv("UGS_STAGE_MAX_MB"); return (int64_t)(e ? std::atoll(e) : 4096) << 20; }();   // read per call (tests toggle it)
    const bool stg = tc.first != UGS_TIER_S && row_count * (int64_t)(UGS_STAGE_ITEMS * sizeof(uint2)) <= stage_max;
    plan->stg_valid = false;
    if (stg) {
        if (int rc = ensure(plan->stage, (size_t)row_count * UGS_STAGE_ITEMS * sizeof(uint2), plan->device, plan)) return rc;
        if (int rc = ensure(plan->ulist, (size_t)row_count * sizeof(int64_t), plan->device, plan)) return rc;
    }
    if (may_overflow || stg) HIP_TRY(hipMemsetAsync(plan->ovfcnt.p, 0, 4 * sizeof(uint32_t), s));   // nothing else reads the counters
    // dynamic work distribution pays when a launch has many more walks than resident groups (its counter costs one memset)
    const bool dyn = row_count > (int64_t)plan->cus * (tc.first == UGS_TIER_S ? 1024 : 64) && std::getenv("UGS_STATIC_SPLIT") == nullptr;
    if (dyn) {
        if (int rc = ensure(plan->work, 4 * sizeof(unsigned long long), plan->device, plan)) return rc;
        HIP_TRY(hipMemsetAsync(plan->work.p, 0, 4 * sizeof(unsigned long long), s));
    }
    if ((tc.first != UGS_TIER_S && tc.first != UGS_TIER_G) || tc.second >= 0)
        if (int rc = ensure_prow(plan, s)) return rc;
    uint32_t *cnt = static_cast<uint32_t *>(plan->ovfcnt.p);
    UgsWalkArgs a{};
    a.plan = plan->dev;
    a.m = m_per_graph; a.k = k; a.mode = mode;
    a.extra_node_off = extra_node_offset;
    a.seed64 = (uint64_t)(int64_t)seed;
    a.seed_ptr = d_seed_ptr;
    a.row_begin = row_begin; a.row_count = row_count;
    a.nodes = d_nodes;
    a.counts = static_cast<uint32_t *>(plan->counts.p);
    a.in_list = nullptr; a.in_count = nullptr;
    a.ovf_list = static_cast<int64_t *>(plan->ovf1.p);
    a.ovf_count = cnt + 0;
    if (stg) {
        a.stage = static_cast<uint2 *>(plan->stage.p);
        a.ulist = static_cast<int64_t *>(plan->ulist.p); a.ucount = cnt + 3;
        plan->stg_valid = true; plan->stg_nodes = d_nodes; plan->stg_row_begin = row_begin; plan->stg_row_count = row_count;
        plan->stg_m = m_per_graph; plan->stg_k = k;
    }
    a.work_next = dyn ? static_cast<unsigned long long *>(plan->work.p) + 0 : nullptr;
    if (tc.small) a.pad = UGS_SMALL_CAP;
    if (tc.wide && !dyn && std::getenv("UGS_NO_WIDE_TIER") == nullptr /* read per call: tests and A/Bs toggle it */ && row_count <= (int64_t)plan->cus * 3 /* blocks per CU, UGS_BLOCKS_S */ * UGS_WIDE_LANES) a.pad |= UGS_WIDE_LANES;       // (UGS_SMALL_CAP | UGS_WIDE_LANES: both)
    // ugs_plan_step: the fill kernel can turn the counts into edge_ptr itself when the walk leaves the sums of 8 rows beside them --
    // 8-lane tier, rows taken by index (static split), no walk handed on, no capture in progress, a row count the fill's blocks can
    // add up in a few dozen loads per thread
    const bool defer = defer_scan && tc.first == UGS_TIER_S && !may_overflow && !dyn && row_count <= 131072 && !capturing(s) &&
                       std::getenv("UGS_NO_FUSED_SCAN") == nullptr;
    if (defer) {
        if (int rc = ensure(plan->tiles, std::max<size_t>((size_t)((row_count + 7) / 8) * sizeof(uint32_t), 4096), plan->device, plan)) return rc;
        a.wsum = static_cast<uint32_t *>(plan->tiles.p);
    }
    HIP_TRY(ev_begin(plan, 0, s));
    HIP_TRY(ugs_launch_walk(a, tc.first, plan->cus, plan->walk_share, s, &plan->last_walk));
    HIP_TRY(ev_end(plan, s));
    HIP_TRY(ev_begin(plan, 1, s));
    int last = 0;   // index of the counter holding rows that nobody processed
    if (tc.second >= 0) {
        a.in_list = static_cast<const int64_t *>(plan->ovf1.p); a.in_count = cnt + 0;
        a.ovf_list = static_cast<int64_t *>(plan->ovf2.p); a.ovf_count = cnt + 1;
        a.work_next = dyn ? static_cast<unsigned long long *>(plan->work.p) + 1 : nullptr;
        HIP_TRY(ugs_launch_walk(a, tc.second, plan->cus, plan->walk_share, s, nullptr));
        last = 1;
    }
    if (tc.third_G) {
        if (!plan->gws.p) {
            int idx = 0;
            const int64_t gcap = std::max<int64_t>(tc.bound, 1);
            const uint32_t gb = ugs_chain_at_least(gcap, &idx);
            if (!gb) return fail(UGS_E_UNSUPPORTED, "candidate-set bound too large");
            const int64_t prev = std::max<int64_t>(ugs_ord_words(idx), 4);            // orders of every stage below the last one
            int64_t hs = 128;
            while (hs < 2 * (gcap + 65 + 1)) hs <<= 1;
            const int64_t words = ugs_global_ws_words(gcap, gb, prev, hs);
            int64_t groups = std::min<int64_t>(256, std::max<int64_t>(1, ((int64_t)4 << 30) / (words * 4)));
            if (words * 4 > ((int64_t)16 << 30)) return fail(UGS_E_UNSUPPORTED, "candidate-set bound too large for the global-memory workspace");
            void *w = nullptr;
            HIP_TRY(hipMalloc(&w, (size_t)(words * groups) * sizeof(uint32_t)));
            plan->gws.p = w; plan->gws.bytes = (size_t)(words * groups) * 4; plan->gws.dev = plan->device;
            plan->gws_groups = groups; plan->gws_words = words;
            plan->gcap = (int)gcap; plan->gbcap = (int)gb; plan->gpcap = (int)prev; plan->ghs = (int)hs;
        }
        a.in_list = static_cast<const int64_t *>(last == 0 ? plan->ovf1.p : plan->ovf2.p); a.in_count = cnt + last;
        a.ovf_list = static_cast<int64_t *>(last == 0 ? plan->ovf2.p : plan->ovf1.p); a.ovf_count = cnt + 2;
        a.gws = static_cast<uint32_t *>(plan->gws.p);
        a.gws_words_per_group = plan->gws_words; a.gws_groups = plan->gws_groups;
        a.gcap = plan->gcap; a.ghs = plan->ghs; a.gbcap = plan->gbcap; a.gpcap = plan->gpcap;
        a.work_next = nullptr;              // the global tier keeps one walk per workspace slice, static
        HIP_TRY(ugs_launch_walk(a, UGS_TIER_G, plan->cus, plan->walk_share, s, nullptr));
        last = 2;
    }
    // the library's own jobs (poll_total): the scan kernel hands the total to the host and the host polls for it -- the caller of the job
    // API gets its outputs from later operations on the same stream, so nothing else needs the stream to be idle here.  Not while rows
    // may have been handed on (the overflow counters are read back with the total) and not under UGS_DEBUG.
    bool polled = false;
    if (poll_total && total_edges_host && !may_overflow && !debug_on() && !capturing(s) && !defer) {
        if (!plan->pin_slot) plan->pin_slot = pin_slot_get();
        polled = plan->pin_slot != nullptr;
    }
    if (defer) *defer_scan = true;
    else if (polled) {
        uint32_t ep = g_pin_epoch.fetch_add(1) + 1;
        if (ep == 0) ep = g_pin_epoch.fetch_add(1) + 1;
        plan->pin_epoch = ep;
        HIP_TRY(ugs_launch_scan(static_cast<const uint32_t *>(plan->counts.p), row_count, d_edge_ptr, static_cast<int64_t *>(plan->scantmp.p), s,
                                reinterpret_cast<int64_t *>(plan->pin_slot), reinterpret_cast<uint32_t *>(plan->pin_slot + 8), ep));
    } else HIP_TRY(ugs_launch_scan(static_cast<const uint32_t *>(plan->counts.p), row_count, d_edge_ptr, static_cast<int64_t *>(plan->scantmp.p), s));
    HIP_TRY(ev_end(plan, s));
    if (int rc = plan_leave(plan, s)) return rc;
    if (defer) return UGS_OK;                   // the caller's fill kernel scans (and, for the library's jobs, reports the total)
    if (polled) {
        HIP_TRY(wait_signal(reinterpret_cast<volatile uint32_t *>(plan->pin_slot + 8), plan->pin_epoch, s));
        *total_edges_host = *reinterpret_cast<volatile int64_t *>(plan->pin_slot);
        plan->last_overflow = 0;
        return UGS_OK;
    }
    if (total_edges_host) {
        uint32_t h[4] = {0, 0, 0, 0};
        int64_t tot = 0;
        HIP_TRY(hipMemcpyAsync(&tot, d_edge_ptr + row_count, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        if (may_overflow) HIP_TRY(hipMemcpyAsync(h, plan->ovfcnt.p, sizeof(h), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        *total_edges_host = tot;
        plan->last_overflow = h[0];
        const bool handled = may_overflow;
        if ((!handled && h[0] != 0) || (handled && h[last] != 0))
            return fail(UGS_E_HIP, "internal error: walk rows left unprocessed after the last tier");
        if (debug_on()) std::fprintf(stderr, "[UGS SAMPLE] rows=%lld k=%d tier=%d overflow=%u/%u/%u edges=%lld\n", (long long)row_count, k, tc.first, h[0], h[1], h[2], (long long)tot);
    }
    return UGS_OK;
}

int ugs_plan_fill(ugs_plan *plan, int m_per_graph, int k, int mode, int64_t extra_node_offset, int64_t row_begin, int64_t row_count,
                  void *stream, const int64_t *d_nodes, const int64_t *d_edge_ptr, int64_t *d_edge_index, int64_t ld, int64_t *d_edge_src) {
    if (!plan) return fail(UGS_E_BAD_ARG, "plan is null");
    if (k < 1 || k > UGS_KMAX) return fail(UGS_E_UNSUPPORTED, "k outside [1, 32]");
    if (mode < 0 || mode > UGS_FILL_BATCH) return fail(UGS_E_BAD_MODE, "mode must be one of: 'sample', 'graph', 'global'");
    if (row_count <= 0) return UGS_OK;
    HIP_TRY(hipSetDevice(plan->device));
    UgsFillArgs a{};
    a.plan = plan->dev;
    a.m = m_per_graph; a.k = k; a.mode = mode;
    a.extra_node_off = extra_node_offset;
    a.row_begin = row_begin; a.row_count = row_count;
    a.nodes = d_nodes; a.edge_ptr = d_edge_ptr;
    a.edge_index = d_edge_index; a.ld = ld; a.edge_src = d_edge_src;
    const TierChoice tc = choose_tier(plan, k);
    std::lock_guard<std::mutex> lk(plan->mu);
    if (plan->stg_valid && plan->stg_nodes == d_nodes && plan->stg_row_begin == row_begin && plan->stg_row_count == row_count &&
        plan->stg_m == m_per_graph && plan->stg_k == k) {
        a.stage = static_cast<const uint2 *>(plan->stage.p); a.counts = static_cast<const uint32_t *>(plan->counts.p);
        a.ulist = static_cast<const int64_t *>(plan->ulist.p); a.ucount = static_cast<const uint32_t *>(plan->ovfcnt.p) + 3;
    }
    if (int rc = plan_enter(plan, static_cast<hipStream_t>(stream))) return rc;
    HIP_TRY(ev_begin(plan, 2, static_cast<hipStream_t>(stream)));
    HIP_TRY(ugs_launch_fill(a, tc.first != UGS_TIER_S, plan->cus, static_cast<hipStream_t>(stream), &plan->last_fill));
    HIP_TRY(ev_end(plan, static_cast<hipStream_t>(stream)));
    return plan_leave(plan, static_cast<hipStream_t>(stream));
}

int ugs_plan_step(ugs_plan *plan, int m_per_graph, int k, int mode, int64_t extra_node_offset, int seed, int64_t row_begin, int64_t row_count,
                  void *stream, int64_t *d_nodes, int64_t *d_edge_ptr, int64_t *d_edge_index, int64_t ld, int64_t *d_edge_src) {
    bool deferred = false;
    if (int rc = plan_walk_impl(plan, m_per_graph, k, mode, extra_node_offset, seed, nullptr, row_begin, row_count, stream, d_nodes, d_edge_ptr,
                                nullptr, &deferred)) return rc;
    if (!deferred || row_count <= 0)
        return ugs_plan_fill(plan, m_per_graph, k, mode, extra_node_offset, row_begin, row_count, stream, d_nodes, d_edge_ptr, d_edge_index, ld, d_edge_src);
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(plan->device));
    std::lock_guard<std::mutex> lk(plan->mu);
    if (int rc = plan_enter(plan, s)) return rc;
    UgsFillArgs a{};
    a.plan = plan->dev;
    a.m = m_per_graph; a.k = k; a.mode = mode;
    a.extra_node_off = extra_node_offset;
    a.row_begin = row_begin; a.row_count = row_count;
    a.nodes = d_nodes; a.edge_ptr = d_edge_ptr; a.edge_ptr_out = d_edge_ptr;
    a.edge_index = d_edge_index; a.ld = ld; a.edge_src = d_edge_src;
    a.counts = static_cast<const uint32_t *>(plan->counts.p);
    a.wsum = static_cast<const uint32_t *>(plan->tiles.p);
    HIP_TRY(ev_begin(plan, 2, s));
    HIP_TRY(ugs_launch_fill_scan(a, plan->cus, s, &plan->last_fill));
    HIP_TRY(ev_end(plan, s));
    return plan_leave(plan, s);
}

// ---- a step captured as a HIP graph (include/ugs_mi355.h) -----------------------------------------------------------------
struct ugs_graph {
    ugs_plan *owner = nullptr;       // keeps the device arrays alive
    ugs_plan *shadow = nullptr;      // the same device arrays with PRIVATE scratch: the graph bakes scratch pointers in, and the
                                     // owner's scratch may be regrown by later calls
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraphNode_t seed_node = nullptr;
    hipStream_t cs = nullptr;
    uint64_t *h_seeds = nullptr;     // pinned ring: a replay's seed must stay put until its upload node has run
    uint64_t *d_seed = nullptr;
    int slot = 0, device = -1;
};
static const int kSeedRing = 256;

int ugs_plan_graph_destroy(ugs_graph *g) {
    if (!g) return UGS_OK;
    if (g->device >= 0 && hipSetDevice(g->device) == hipSuccess) (void)hipDeviceSynchronize();
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->cs) (void)hipStreamDestroy(g->cs);
    if (g->h_seeds) (void)hipHostFree(g->h_seeds);
    if (g->d_seed) (void)hipFree(g->d_seed);
    if (g->shadow) plan_unref(g->shadow);
    if (g->owner) plan_unref(g->owner);
    delete g;
    return UGS_OK;
}

int ugs_plan_graph_create(ugs_plan *plan, int m_per_graph, int k, int mode, int64_t extra_node_offset, int64_t row_begin, int64_t row_count,
                          int64_t *d_nodes, int64_t *d_edge_ptr, int64_t *d_edge_index, int64_t ld, int64_t *d_edge_src, ugs_graph **graph_out) {
    if (!plan || !graph_out) return fail(UGS_E_BAD_ARG, "null argument");
    if (row_count <= 0 || !d_nodes || !d_edge_ptr || !d_edge_index || !d_edge_src || ld <= 0) return fail(UGS_E_BAD_ARG, "a captured step needs rows and all four output buffers");
    HIP_TRY(hipSetDevice(plan->device));
    auto *g = new ugs_graph();
    g->device = plan->device;
    plan->refs.fetch_add(1);
    g->owner = plan;
    {   // the padded rows belong to the OWNER (one copy, counted against the plan cache): build them before the shadow copies `dev`
        const TierChoice tc0 = choose_tier(plan, k);
        std::lock_guard<std::mutex> lk(plan->mu);
        if ((tc0.first != UGS_TIER_S && tc0.first != UGS_TIER_G) || tc0.second >= 0) {
            DeviceCtx dc0;
            if (int rc = device_ctx(dc0)) { plan_unref(plan); delete g; return rc; }
            if (int rc = ensure_prow(plan, dc0.stream)) { plan_unref(plan); delete g; return rc; }
        }
    }
    auto *sh = new ugs_plan();
    sh->device = plan->device; sh->cus = plan->cus; sh->G = plan->G; sh->nverts = plan->nverts; sh->nnz = plan->nnz;
    {
        std::lock_guard<std::mutex> lk(plan->mu);
        sh->dev = plan->dev;         // device arrays shared, not owned (blob stays null)
    }
    sh->prow_failed = true;          // never a private copy of the padded rows: the owner's, or the row-pointer path
    sh->g_n = plan->g_n; sh->g_maxdeg = plan->g_maxdeg; sh->g_sbdeg = plan->g_sbdeg; sh->g_level = plan->g_level;
    sh->walk_share = plan->walk_share;
    g->shadow = sh;
    auto bail = [&](int rc) { ugs_plan_graph_destroy(g); return rc; };
    hipError_t e = hipStreamCreateWithFlags(&g->cs, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&g->h_seeds), kSeedRing * sizeof(uint64_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&g->d_seed), sizeof(uint64_t));
    if (e != hipSuccess) return bail(fail_hip(e, "graph resources"));
    g->h_seeds[0] = 0;
    // 1. an ordinary step: every scratch buffer of the shadow plan gets its final size before anything is captured
    if (int rc = plan_walk_impl(sh, m_per_graph, k, mode, extra_node_offset, 0, nullptr, row_begin, row_count, g->cs, d_nodes, d_edge_ptr, nullptr)) return bail(rc);
    if (int rc = ugs_plan_fill(sh, m_per_graph, k, mode, extra_node_offset, row_begin, row_count, g->cs, d_nodes, d_edge_ptr, d_edge_index, ld, d_edge_src)) return bail(rc);
    e = hipStreamSynchronize(g->cs);
    if (e != hipSuccess) return bail(fail_hip(e, "graph warm-up"));
    // 2. the same step captured, reading its seed from d_seed
    e = hipStreamBeginCapture(g->cs, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return bail(fail_hip(e, "hipStreamBeginCapture"));
    int rc = plan_walk_impl(sh, m_per_graph, k, mode, extra_node_offset, 0, g->d_seed, row_begin, row_count, g->cs, d_nodes, d_edge_ptr, nullptr);
    if (!rc) rc = ugs_plan_fill(sh, m_per_graph, k, mode, extra_node_offset, row_begin, row_count, g->cs, d_nodes, d_edge_ptr, d_edge_index, ld, d_edge_src);
    e = hipStreamEndCapture(g->cs, &g->graph);
    if (rc) return bail(rc);
    if (e != hipSuccess || !g->graph) return bail(fail_hip(e, "hipStreamEndCapture"));
    // 3. the seed upload as an explicit root node every captured root depends on (its source slot is re-pointed per replay)
    size_t nroots = 0;
    e = hipGraphGetRootNodes(g->graph, nullptr, &nroots);
    std::vector<hipGraphNode_t> roots(nroots);
    if (e == hipSuccess && nroots) e = hipGraphGetRootNodes(g->graph, roots.data(), &nroots);
    const bool static_seed = std::getenv("UGS_GRAPH_STATIC") != nullptr;   // measurement aid: replay without the seed upload
    if (e == hipSuccess && !static_seed) e = hipGraphAddMemcpyNode1D(&g->seed_node, g->graph, nullptr, 0, g->d_seed, g->h_seeds, sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && nroots && !static_seed) {
        std::vector<hipGraphNode_t> from(nroots, g->seed_node);
        e = hipGraphAddDependencies(g->graph, from.data(), roots.data(), nroots);
    }
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) return bail(fail_hip(e, "building the graph"));
    *graph_out = g;
    return UGS_OK;
}

// A second plan over the SAME device arrays with scratch of its own: calls on a plan are serialised through its scratch (one
// stream at a time), so a caller that wants two steps in flight -- the next batch's walk filling the tail of this one's, on
// another stream -- samples them through a plan and its twin alternately.
int ugs_plan_twin(ugs_plan *plan, int k, ugs_plan **twin_out) {
    if (!plan || !twin_out) return fail(UGS_E_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(plan->device));
    {   // the padded rows belong to the owner: build them before the twin copies `dev`
        const TierChoice tc0 = choose_tier(plan, k);
        std::lock_guard<std::mutex> lk(plan->mu);
        if ((tc0.first != UGS_TIER_S && tc0.first != UGS_TIER_G) || tc0.second >= 0) {
            DeviceCtx dc0;
            if (int rc = device_ctx(dc0)) return rc;
            if (int rc = ensure_prow(plan, dc0.stream)) return rc;
        }
    }
    auto *sh = new ugs_plan();
    sh->device = plan->device; sh->cus = plan->cus; sh->G = plan->G; sh->nverts = plan->nverts; sh->nnz = plan->nnz;
    {
        std::lock_guard<std::mutex> lk(plan->mu);
        // The twin shares the owner's device arrays but not its ordering state.  A device-built plan's graph descriptors go up by a
        // truly asynchronous copy whose only guard is the owner's recorded event (plan_leave): whatever the owner still has in
        // flight is waited for here, once, so that the twin's first walk on any stream reads finished arrays.
        if (plan->last_valid) {
            const hipError_t e = hipEventSynchronize(plan->last_ev);
            if (e != hipSuccess) { delete sh; return fail(UGS_E_HIP, hipGetErrorString(e)); }
        }
        sh->dev = plan->dev;
        sh->walk_share = plan->walk_share;
    }
    sh->g_n = plan->g_n; sh->g_maxdeg = plan->g_maxdeg; sh->g_sbdeg = plan->g_sbdeg; sh->g_level = plan->g_level;
    sh->handle_api = plan->handle_api;
    sh->prow_failed = true;
    plan->refs.fetch_add(1);
    sh->twin_of = plan;
    *twin_out = sh;
    return UGS_OK;
}

int ugs_plan_graph_launch(ugs_graph *g, int seed, void *stream) {
    if (!g || !g->exec) return fail(UGS_E_BAD_ARG, "graph is null");
    HIP_TRY(hipSetDevice(g->device));
    g->slot = (g->slot + 1) % kSeedRing;
    g->h_seeds[g->slot] = (uint64_t)(int64_t)seed;
    if (g->seed_node) HIP_TRY(hipGraphExecMemcpyNodeSetParams1D(g->exec, g->seed_node, g->d_seed, g->h_seeds + g->slot, sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream)));
    return UGS_OK;
}

// ---- collation of a sharded batch (multi-GPU path; include/ugs_mi355.h: wire format) -------------------------------------------
int ugs_collate_layout(int k, int node_bytes, int eidx_bytes, int esrc_bytes, int64_t rows_cap, int64_t edge_cap, int64_t *section_off4,
                       int64_t *msg_bytes) {
    if (k < 1 || rows_cap < 0 || edge_cap < 0 || !section_off4 || !msg_bytes) return fail(UGS_E_BAD_ARG, "bad arguments to collate_layout");
    if ((node_bytes != 4 && node_bytes != 8) || (eidx_bytes != 1 && eidx_bytes != 4 && eidx_bytes != 8) || (esrc_bytes != 4 && esrc_bytes != 8))
        return fail(UGS_E_BAD_ARG, "wire widths: nodes 4|8, edge_index 1|4|8, edge_src 4|8 bytes");
    auto up = [](int64_t x) { return (x + 15) & ~(int64_t)15; };
    int64_t off = 16;                                            // header: rows, edge entries (2 x int64)
    section_off4[0] = off; off = up(off + rows_cap * k * node_bytes);
    section_off4[1] = off; off = up(off + (rows_cap + 1) * 4);
    section_off4[2] = off; off = up(off + 2 * edge_cap * eidx_bytes);
    section_off4[3] = off; off = up(off + edge_cap * esrc_bytes);
    *msg_bytes = off;
    return UGS_OK;
}

int ugs_collate_unpack(const void *d_msgs, int world, const int64_t *row_off, int k, int node_bytes, int eidx_bytes, int esrc_bytes,
                       int64_t rows_cap, int64_t edge_cap, int64_t *d_nodes, int64_t *d_edge_index, int64_t ld, int64_t *d_edge_ptr,
                       int64_t *d_edge_src, int64_t *d_max_total, void *stream) {
    if (!d_msgs || !row_off || world < 1 || world > UGS_COLLATE_MAX_WORLD) return fail(UGS_E_BAD_ARG, "collate: 1 <= world <= 64 messages expected");
    if (!d_nodes || !d_edge_ptr || (edge_cap > 0 && (!d_edge_index || !d_edge_src))) return fail(UGS_E_BAD_ARG, "null output pointer");
    int64_t so[4], mb = 0;
    if (int rc = ugs_collate_layout(k, node_bytes, eidx_bytes, esrc_bytes, rows_cap, edge_cap, so, &mb)) return rc;
    for (int r = 0; r < world; ++r)
        if (row_off[r + 1] < row_off[r] || row_off[r + 1] - row_off[r] > rows_cap) return fail(UGS_E_BAD_ARG, "collate: a rank's row range exceeds rows_cap");
    HIP_TRY(ugs_launch_collate_unpack(d_msgs, world, mb, row_off, k, node_bytes, eidx_bytes, esrc_bytes, rows_cap, edge_cap, so, d_nodes,
                                      d_edge_index, ld, d_edge_ptr, d_edge_src, d_max_total, static_cast<hipStream_t>(stream)));
    return UGS_OK;
}

int ugs_plan_set_walk_share(ugs_plan *plan, int percent) {
    if (!plan || percent < 1 || percent > 100) return fail(UGS_E_BAD_ARG, "walk share: a plan and 1..100 percent");
    std::lock_guard<std::mutex> lk(plan->mu);
    plan->walk_share = percent;
    return UGS_OK;
}

int ugs_plan_set_timing(ugs_plan *plan, int on) {
    if (!plan) return fail(UGS_E_BAD_ARG, "plan is null");
    std::lock_guard<std::mutex> lk(plan->mu);
    ev_clear(plan);
    plan->timing = on != 0;
    return UGS_OK;
}

int ugs_plan_get_timing(ugs_plan *plan, double *ms_sum3, int64_t *launches3) {
    if (!plan || !ms_sum3 || !launches3) return fail(UGS_E_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(plan->mu);
    for (int i = 0; i < 3; ++i) { ms_sum3[i] = 0.0; launches3[i] = 0; }
    for (auto &e : plan->events) {
        HIP_TRY(hipEventSynchronize(e.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
        ms_sum3[e.kind] += (double)ms;
        launches3[e.kind] += 1;
    }
    ev_clear(plan);
    return UGS_OK;
}

int ugs_plan_last_launch(const ugs_plan *plan, char *name_buf, int name_buf_len, int *grid, int *block, int *lds_bytes, int64_t *overflow_rows) {
    if (!plan) return fail(UGS_E_BAD_ARG, "plan is null");
    if (name_buf && name_buf_len > 0) { std::snprintf(name_buf, (size_t)name_buf_len, "%s", plan->last_walk.name ? plan->last_walk.name : ""); }
    if (grid) *grid = plan->last_walk.grid;
    if (block) *block = plan->last_walk.block;
    if (lds_bytes) *lds_bytes = plan->last_walk.lds_bytes;
    if (overflow_rows) *overflow_rows = plan->last_overflow;
    return UGS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// jobs: the two-phase host-facing API (begin = walks + counts, finish = fill + copy-out)
// ---------------------------------------------------------------------------------------------------------------
struct ugs_job {
    ugs_plan *plan = nullptr;
    DeviceCtx dc;
    int m = 0, k = 0, mode = 0;
    int64_t extra = 0, rows = 0, total = 0, G = 0;
    bool batch = false;
    PoolBuf nodes;                     // nodes [rows, k] and, right behind it, edge_ptr [rows + 1]: one buffer, so that a caller whose
    int64_t *d_eptr = nullptr;         // output tensors are adjacent too gets both with one copy
    // batches of small graphs: begin ran walk + fill as one step (scan folded into the fill) into this staging -- edge_index [2, total]
    // and edge_src [total] laid out for the total the kernel found -- and finish only copies out
    bool packed_ok = false;
    // epsilon_uniform path
    bool eps = false;
    PoolBuf eps_blob;                  // pooled: hipMalloc/hipFree per call cost milliseconds once the process holds large plans
    UgsEpsLaunch eps_l{};
    PoolBuf eps_counts, eps_scantmp;
};

namespace {
std::atomic<int64_t> g_spec_kept{0}, g_spec_wrong{0};              // large calls whose early start stood / was thrown away
void free_job(ugs_job *j) {
    if (!j) return;
    pool_put(j->nodes);
    pool_put(j->eps_counts); pool_put(j->eps_scantmp);
    pool_put(j->eps_blob);
    plan_unref(j->plan);
    delete j;
}

// second launch of a job's step (see begin_common): ugs_fill_scan into the job's staging, total through the pinned slot of the plan
int packed_fill(ugs_job *j, int64_t cap3) {
    ugs_plan *plan = j->plan;
    hipStream_t s = j->dc.stream;
    int64_t *stg = static_cast<int64_t *>(j->nodes.p) + (j->rows * j->k + j->rows + 1);       // (begin_common sized the buffer for it)
    std::unique_lock<std::mutex> lk(plan->mu);
    if (int rc = plan_enter(plan, s)) return rc;
    if (!plan->pin_slot) plan->pin_slot = pin_slot_get();
    UgsFillArgs a{};
    a.plan = plan->dev;
    a.m = j->m; a.k = j->k; a.mode = j->mode;
    a.extra_node_off = j->extra;
    a.row_begin = 0; a.row_count = j->rows;
    a.nodes = static_cast<const int64_t *>(j->nodes.p); a.edge_ptr = j->d_eptr; a.edge_ptr_out = j->d_eptr;
    a.edge_index = stg; a.ld = 0; a.edge_src = nullptr;                                       // set by the kernel (packed_cap)
    a.packed_cap = cap3;
    a.counts = static_cast<const uint32_t *>(plan->counts.p);
    a.wsum = static_cast<const uint32_t *>(plan->tiles.p);
    const bool poll = plan->pin_slot != nullptr;
    if (poll) {
        uint32_t ep = g_pin_epoch.fetch_add(1) + 1;
        if (ep == 0) ep = g_pin_epoch.fetch_add(1) + 1;
        plan->pin_epoch = ep;
        a.h_total = reinterpret_cast<int64_t *>(plan->pin_slot); a.h_flag = reinterpret_cast<uint32_t *>(plan->pin_slot + 8); a.epoch = ep;
    }
    HIP_TRY(ev_begin(plan, 2, s));
    HIP_TRY(ugs_launch_fill_scan(a, plan->cus, s, &plan->last_fill));
    HIP_TRY(ev_end(plan, s));
    if (int rc = plan_leave(plan, s)) return rc;
    if (poll) {
        HIP_TRY(wait_signal(reinterpret_cast<volatile uint32_t *>(plan->pin_slot + 8), plan->pin_epoch, s));
        j->total = *reinterpret_cast<volatile int64_t *>(plan->pin_slot);
    } else {
        int64_t tot = 0;
        HIP_TRY(hipMemcpyAsync(&tot, j->d_eptr + j->rows, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        j->total = tot;
    }
    plan->last_overflow = 0;
    j->packed_ok = 3 * j->total <= cap3;
    return UGS_OK;
}

int begin_common(ugs_plan *plan, int m, int k, int mode, int64_t extra, int seed, bool batch, ugs_job **job_out, int64_t *total_out) {
    DeviceCtx dc;
    if (int rc = device_ctx(dc)) { plan_unref(plan); return rc; }
    auto *j = new ugs_job();
    j->plan = plan; j->dc = dc; j->m = m; j->k = k; j->mode = mode; j->extra = extra; j->batch = batch;
    j->G = plan->G;
    j->rows = plan->G * (int64_t)m;
    // (a job that may run the packed step keeps its edge staging right behind nodes and edge_ptr: all four outputs leave in ONE copy when
    // the caller's tensors are adjacent too, as the Python shim's are)
    const int64_t node_words = j->rows * k + j->rows + 1;
    const int64_t cap3_want = 3 * j->rows * 2 * (int64_t)k * (int64_t)(k - 1);
    const TierChoice tc0 = choose_tier(plan, k);
    const bool may_pack = k >= 2 && j->rows > 0 && j->rows <= 131072 && tc0.first == UGS_TIER_S && tc0.second < 0 && !tc0.third_G &&
                          cap3_want * (int64_t)sizeof(int64_t) <= ((int64_t)192 << 20) && !debug_on() && std::getenv("UGS_NO_PACKED_STEP") == nullptr;
    int rc = pool_get((size_t)(node_words + (may_pack ? cap3_want : 0)) * sizeof(int64_t), dc.id, j->nodes);
    if (!rc) {
        j->d_eptr = static_cast<int64_t *>(j->nodes.p) + j->rows * k;
        // Small batches: every ordered pair of a row's vertices, twice (both directions of a PyG edge are columns, and each column is
        // symmetrised), bounds the edge entries -- if a staging of that size is affordable the step runs here in two launches, the total
        // comes from the fill kernel's first block, and the caller allocates while the rows are being filled.  Repeated columns can
        // exceed the bound: the kernel then writes nothing and finish fills the ordinary way.
        const int64_t cap3 = cap3_want;
        const bool want_packed = may_pack;
        bool deferred = false;
        rc = plan_walk_impl(plan, m, k, mode, extra, seed, nullptr, 0, j->rows, dc.stream, static_cast<int64_t *>(j->nodes.p), j->d_eptr, &j->total,
                            want_packed ? &deferred : nullptr, true);
        if (!rc && deferred) rc = packed_fill(j, cap3);
    }
    if (rc) { free_job(j); return rc; }
    *job_out = j;
    if (total_out) *total_out = j->total;
    return UGS_OK;
}

int finish_common(ugs_job *j, int64_t *nodes, int64_t *edge_index, int64_t *edge_ptr, int64_t *sample_ptr, int64_t *edge_src, int dst_is_device) {
    hipStream_t s = j->dc.stream;
    const int64_t rows = j->rows, k = j->k, tot = j->total;
    int rc = UGS_OK;
    PoolBuf e_idx;                     // host outputs: edge_index [2, tot] and edge_src [tot] staged in ONE device buffer
    auto body = [&]() -> int {
        HIP_TRY(hipSetDevice(j->dc.id));
        int64_t *d_ei = edge_index, *d_es = edge_src;
        const bool packed = j->packed_ok && tot > 0;                // the step already ran (begin_common): copy out of its staging
        if (packed) { d_ei = static_cast<int64_t *>(j->nodes.p) + (rows * k + rows + 1); d_es = d_ei + 2 * tot; }
        else if (!dst_is_device && tot > 0) {
            if (int r = pool_get((size_t)(3 * tot) * sizeof(int64_t), j->dc.id, e_idx)) return r;
            d_ei = static_cast<int64_t *>(e_idx.p); d_es = d_ei + 2 * tot;
        }
        if (tot > 0 && j->eps) {
            if (!d_ei || !d_es) return fail(UGS_E_BAD_ARG, "null edge output pointer");
            UgsEpsLaunch l = j->eps_l;
            l.edge_ptr = j->d_eptr; l.edge_index = d_ei; l.edge_src = d_es; l.ld = tot;
            HIP_TRY(ugs_eps_launch(l, 1, j->dc.cus, s));
        } else if (tot > 0 && !packed) {
            if (!d_ei || !d_es) return fail(UGS_E_BAD_ARG, "null edge output pointer");
            if (int r = ugs_plan_fill(j->plan, j->m, j->k, j->mode, j->extra, 0, rows, s, static_cast<const int64_t *>(j->nodes.p),
                                      j->d_eptr, d_ei, tot, d_es)) return r;
        }
        const hipMemcpyKind kind = dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        // adjacent outputs (the Python shim carves its tensors out of one allocation) travel in one copy each -- or all in one
        if (packed && edge_ptr == nodes + rows * k && edge_index == edge_ptr + rows + 1 && edge_src == edge_index + 2 * tot) {
            // (the same copy through a kernel storing to the pinned tensors was measured: 103 against 96 us for the 2.3 MB of the
            // PROTEINS-shaped call -- the DMA copy stays)
            HIP_TRY(hipMemcpyAsync(nodes, j->nodes.p, (size_t)(rows * k + rows + 1 + 3 * tot) * sizeof(int64_t), kind, s));
        } else {
        if (edge_ptr == nodes + rows * k) {
            HIP_TRY(hipMemcpyAsync(nodes, j->nodes.p, (size_t)(rows * k + rows + 1) * sizeof(int64_t), kind, s));
        } else {
            if (rows * k > 0) HIP_TRY(hipMemcpyAsync(nodes, j->nodes.p, (size_t)(rows * k) * sizeof(int64_t), kind, s));
            HIP_TRY(hipMemcpyAsync(edge_ptr, j->d_eptr, (size_t)(rows + 1) * sizeof(int64_t), kind, s));
        }
        if ((!dst_is_device || packed) && tot > 0) {
            if (!edge_index || !edge_src) return fail(UGS_E_BAD_ARG, "null edge output pointer");
            if (edge_src == edge_index + 2 * tot) {
                HIP_TRY(hipMemcpyAsync(edge_index, d_ei, (size_t)(3 * tot) * sizeof(int64_t), kind, s));
            } else {
                HIP_TRY(hipMemcpyAsync(edge_index, d_ei, (size_t)(2 * tot) * sizeof(int64_t), kind, s));
                HIP_TRY(hipMemcpyAsync(edge_src, d_es, (size_t)tot * sizeof(int64_t), kind, s));
            }
        }
        }
        if (sample_ptr) {
            std::vector<int64_t> sp((size_t)j->G + 1);
            for (int64_t g = 0; g <= j->G; ++g) sp[(size_t)g] = g * (int64_t)j->m;
            if (dst_is_device) HIP_TRY(hipMemcpy(sample_ptr, sp.data(), sp.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            else std::memcpy(sample_ptr, sp.data(), sp.size() * sizeof(int64_t));
        }
        HIP_TRY(hipStreamSynchronize(s));
        return UGS_OK;
    };
    rc = body();
    pool_put(e_idx);
    free_job(j);
    return rc;
}
}  // namespace

extern "C" {

int ugs_sample_begin(int64_t handle, int m_per_graph, int k, int edge_mode, int64_t base_offset, int seed, ugs_job **job_out, int64_t *total_edges_out) {
    if (!job_out) return fail(UGS_E_BAD_ARG, "job_out is null");
    if (edge_mode < 0 || edge_mode > 2) return fail(UGS_E_BAD_MODE, "edge_mode must be one of: 'local', 'flat', 'global'");
    if (m_per_graph < 0) return fail(UGS_E_BAD_ARG, "m_per_graph must be >= 0");
    auto g = lookup(handle);
    if (!g) return fail(UGS_E_INVALID_HANDLE, "Invalid preproc handle");
    if (g->n == 0) return fail(UGS_E_NO_ROOTS, "No viable roots available");
    ugs_plan *plan = nullptr;
    if (int rc = ugs_plan_create_handle(handle, &plan)) return rc;
    return begin_common(plan, m_per_graph, k, edge_mode, edge_mode == UGS_EDGE_GLOBAL ? base_offset : 0, seed, false, job_out, total_edges_out);
}

int ugs_sample_finish(ugs_job *job, int64_t *nodes, int64_t *edge_index, int64_t *edge_ptr, int64_t *edge_src, int dst_is_device) {
    if (!job) return fail(UGS_E_BAD_ARG, "job is null");
    return finish_common(job, nodes, edge_index, edge_ptr, nullptr, edge_src, dst_is_device);
}

int ugs_sample_batch_begin(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr, int64_t num_graphs,
                           int m_per_graph, int k, int mode, int seed, ugs_job **job_out, int64_t *total_edges_out) {
    if (!job_out) return fail(UGS_E_BAD_ARG, "job_out is null");
    if (mode < 0 || mode > 2) return fail(UGS_E_BAD_MODE, "mode must be one of: 'sample', 'graph', 'global'");
    if (m_per_graph < 0) return fail(UGS_E_BAD_ARG, "m_per_graph must be >= 0");
    if (k < 1) return fail(UGS_E_BAD_ARG, "k must be >= 1");
    if (k > UGS_KMAX) return fail(UGS_E_UNSUPPORTED, "k > 32 is not supported by the HIP sampler");
    ugs_plan *plan = nullptr;
    // Large batches start early, like ugs_sample_batch_stream below: the walks begin on the plan the batch's sampled words point at while
    // the lookup (content hash over every column: ~3 ms for 20 M columns) runs on a helper thread; kept only if it names the same plan.
    ugs_plan *guess = nullptr;
    DeviceCtx dc;
    const int64_t spec_min_cols = [] { const char *e = std::getenv("UGS_SPEC_MIN_COLS"); return e ? (int64_t)std::atoll(e) : (int64_t)1 << 21; }();   // (tests lower it)
    if (num_cols >= spec_min_cols && ptr && edge_index && num_graphs > 0 && m_per_graph > 0 && !debug_on() &&
        std::getenv("UGS_NO_SPECULATION") == nullptr && std::getenv("UGS_NO_BATCH_INDEX") == nullptr && device_ctx(dc) == UGS_OK)
        guess = peek_batch_plan(edge_index, edge_index + row_stride, num_cols, ptr, num_graphs, k, dc.id);
    if (guess) {
        int lookup_rc = UGS_OK;
        std::string lookup_err;
        const int dev_tl = t_device; const hipStream_t st_tl = t_job_stream; const bool set_tl = t_job_stream_set;
        std::thread lookup_thread;
        try {
            lookup_thread = std::thread([&, dev_tl, st_tl, set_tl] {
                t_device = dev_tl >= 0 ? dev_tl : dc.id; t_job_stream = st_tl; t_job_stream_set = set_tl;
                lookup_rc = ugs_plan_create_batch(edge_index, row_stride, num_cols, ptr, num_graphs, k, &plan);
                if (lookup_rc != UGS_OK) lookup_err = t_err;
            });
        } catch (...) { plan_unref(guess); guess = nullptr; }          // no thread to be had: look up first, as without a guess
        if (!guess) {
            if (int rc = ugs_plan_create_batch(edge_index, row_stride, num_cols, ptr, num_graphs, k, &plan)) return rc;
            return begin_common(plan, m_per_graph, k, mode, 0, seed, true, job_out, total_edges_out);
        }
        *job_out = nullptr;
        const int rc = begin_common(guess, m_per_graph, k, mode, 0, seed, true, job_out, total_edges_out);   // (owns the guess's reference)
        lookup_thread.join();
        if (lookup_rc != UGS_OK) { if (rc == UGS_OK) { free_job(*job_out); *job_out = nullptr; } return fail(lookup_rc, lookup_err); }
        if (plan == guess) { plan_unref(plan); g_spec_kept.fetch_add(1); return rc; }
        if (rc == UGS_OK) { free_job(*job_out); *job_out = nullptr; }                       // a different batch after all: once more, on its plan
        g_spec_wrong.fetch_add(1);
        return begin_common(plan, m_per_graph, k, mode, 0, seed, true, job_out, total_edges_out);
    }
    if (int rc = ugs_plan_create_batch(edge_index, row_stride, num_cols, ptr, num_graphs, k, &plan)) return rc;
    return begin_common(plan, m_per_graph, k, mode, 0, seed, true, job_out, total_edges_out);
}

int ugs_sample_batch_finish(ugs_job *job, int64_t *nodes, int64_t *edge_index, int64_t *edge_ptr, int64_t *sample_ptr,
                            int64_t *edge_src_global, int dst_is_device) {
    if (!job) return fail(UGS_E_BAD_ARG, "job is null");
    return finish_common(job, nodes, edge_index, edge_ptr, sample_ptr, edge_src_global, dst_is_device);
}

int ugs_job_cancel(ugs_job *job) { free_job(job); return UGS_OK; }

// ---- the whole sample_batch call with the copy-out running BESIDE the walks (large host-visible calls) ------------------------------
// The two-phase call is walk (all rows) -> total -> caller allocates -> fill -> copy out: for a million rows the 400 MB of int64
// results cross PCIe for longer than the walks take, one after the other.  Here the caller hands over its buffers up front (edge
// buffers by a capacity it expects, e.g. the previous call's total plus a margin), the rows go through walk / scan / fill in chunks on
// the job stream, and every finished chunk leaves on a second stream while the next one walks.  Row 1 of edge_index [2, total] starts
// at `total`, known with the last chunk only: that half is kept in the device staging and leaves at the end.
namespace {
std::mutex g_copy_mu;
std::map<int, hipStream_t> &g_copy_streams = *new std::map<int, hipStream_t>();
int copy_stream(int dev, hipStream_t &out) {
    std::lock_guard<std::mutex> lk(g_copy_mu);
    auto it = g_copy_streams.find(dev);
    if (it == g_copy_streams.end()) {
        hipStream_t c = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        it = g_copy_streams.emplace(dev, c).first;
    }
    out = it->second;
    return UGS_OK;
}

// One streamed call: parameters, the caller's (pinned host) buffers, and the device staging that outlives a thrown-away early start.
struct StreamCall {
    DeviceCtx dc;
    int64_t G = 0, extra = 0, cap = 0;
    int m = 0, k = 0, mode = 0, seed = 0;
    int64_t *nodes = nullptr, *edge_index = nullptr, *edge_ptr = nullptr, *sample_ptr = nullptr, *edge_src = nullptr, *total_out = nullptr;
    hipStream_t cs = nullptr;
    PoolBuf d_nodes_b, d_eptr_b, d_loc_b, d_edges_b;
    std::vector<hipEvent_t> evs;
    void release() {
        for (hipEvent_t ev : evs) (void)hipEventDestroy(ev);
        evs.clear();
        pool_put(d_nodes_b); pool_put(d_eptr_b); pool_put(d_loc_b); pool_put(d_edges_b);
    }
};

int stream_rows(ugs_plan *plan, StreamCall &c) {
    const int64_t rows = c.G * (int64_t)c.m, cap = c.cap;
    const int k = c.k;
    int64_t chunk = (rows + 7) / 8;                                   // eight chunks: the first copy starts after an eighth of the walks
    if (chunk < 65536) chunk = 65536;
    if (const char *e = std::getenv("UGS_STREAM_CHUNK_ROWS")) { const int64_t v = std::atoll(e); if (v > 0) chunk = v; }   // (tests: many chunks of a small call)
    const int64_t nchunks = rows > 0 ? (rows + chunk - 1) / chunk : 0;
    hipStream_t s = c.dc.stream;
    int64_t base = 0;
    HIP_TRY(hipSetDevice(c.dc.id));
    if (int rc = copy_stream(c.dc.id, c.cs)) return rc;
    hipStream_t cs = c.cs;
    if (c.sample_ptr) for (int64_t g = 0; g <= c.G; ++g) c.sample_ptr[g] = g * (int64_t)c.m;
    if (rows == 0) { c.edge_ptr[0] = 0; *c.total_out = 0; return UGS_OK; }
    if (!c.d_nodes_b.p) if (int rc = pool_get((size_t)(rows * k) * sizeof(int64_t), c.dc.id, c.d_nodes_b)) return rc;
    if (!c.d_eptr_b.p) if (int rc = pool_get((size_t)(rows + 1) * sizeof(int64_t), c.dc.id, c.d_eptr_b)) return rc;
    if (!c.d_loc_b.p) if (int rc = pool_get((size_t)(chunk + 1) * sizeof(int64_t), c.dc.id, c.d_loc_b)) return rc;
    if (cap > 0 && !c.d_edges_b.p) if (int rc = pool_get((size_t)(3 * cap) * sizeof(int64_t), c.dc.id, c.d_edges_b)) return rc;
    int64_t *d_nodes = static_cast<int64_t *>(c.d_nodes_b.p), *d_eptr = static_cast<int64_t *>(c.d_eptr_b.p), *d_loc = static_cast<int64_t *>(c.d_loc_b.p);
    int64_t *d_ei = static_cast<int64_t *>(c.d_edges_b.p), *d_es = d_ei ? d_ei + 2 * cap : nullptr;
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int64_t r0 = ch * chunk, rc_rows = std::min(chunk, rows - r0);
        const bool last = ch + 1 == nchunks;
        int64_t tot = 0;
        if (int rc = plan_walk_impl(plan, c.m, k, c.mode, c.extra, c.seed, nullptr, r0, rc_rows, s, d_nodes + r0 * k, d_loc, &tot, nullptr, true)) return rc;
        if (base + tot > cap) { *c.total_out = base + tot; return fail(UGS_E_CAPACITY, "edge_capacity too small for this call's edge entries"); }
        if (tot > 0)
            if (int rc = ugs_plan_fill(plan, c.m, k, c.mode, c.extra, r0, rc_rows, s, d_nodes + r0 * k, d_loc, d_ei + base, cap, d_es + base)) return rc;
        HIP_TRY(ugs_launch_rebase_edge_ptr(d_loc, d_eptr + r0, rc_rows + (last ? 1 : 0), base, s));
        hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c.evs.push_back(ev);
        HIP_TRY(hipEventRecord(ev, s));
        HIP_TRY(hipStreamWaitEvent(cs, ev, 0));
        HIP_TRY(hipMemcpyAsync(c.nodes + r0 * k, d_nodes + r0 * k, (size_t)(rc_rows * k) * sizeof(int64_t), hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipMemcpyAsync(c.edge_ptr + r0, d_eptr + r0, (size_t)(rc_rows + (last ? 1 : 0)) * sizeof(int64_t), hipMemcpyDeviceToHost, cs));
        if (tot > 0) {
            HIP_TRY(hipMemcpyAsync(c.edge_index + base, d_ei + base, (size_t)tot * sizeof(int64_t), hipMemcpyDeviceToHost, cs));
            HIP_TRY(hipMemcpyAsync(c.edge_src + base, d_es + base, (size_t)tot * sizeof(int64_t), hipMemcpyDeviceToHost, cs));
        }
        base += tot;
    }
    // row 1 of edge_index: its place in the caller's [2, total] is known now (the last chunk's event orders it behind every fill)
    if (base > 0) HIP_TRY(hipMemcpyAsync(c.edge_index + base, d_ei + cap, (size_t)base * sizeof(int64_t), hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipStreamSynchronize(cs));
    *c.total_out = base;
    return UGS_OK;
}
}  // namespace

int ugs_stream_stats(int64_t *early_starts_kept, int64_t *early_starts_discarded) {
    if (early_starts_kept) *early_starts_kept = g_spec_kept.load();
    if (early_starts_discarded) *early_starts_discarded = g_spec_wrong.load();
    return UGS_OK;
}

int ugs_sample_batch_stream(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr, int64_t num_graphs,
                            int m_per_graph, int k, int mode, int seed, int64_t edge_capacity, int64_t *nodes, int64_t *edge_index_out,
                            int64_t *edge_ptr, int64_t *sample_ptr, int64_t *edge_src_global, int64_t *total_edges_out) {
    if (mode < 0 || mode > 2) return fail(UGS_E_BAD_MODE, "mode must be one of: 'sample', 'graph', 'global'");
    if (m_per_graph < 0) return fail(UGS_E_BAD_ARG, "m_per_graph must be >= 0");
    if (k < 1) return fail(UGS_E_BAD_ARG, "k must be >= 1");
    if (k > UGS_KMAX) return fail(UGS_E_UNSUPPORTED, "k > 32 is not supported by the HIP sampler");
    if (edge_capacity < 0 || !edge_ptr || !total_edges_out) return fail(UGS_E_BAD_ARG, "null output pointer or negative edge_capacity");
    if (!ptr || num_graphs < 0 || num_cols < 0 || (num_cols > 0 && !edge_index)) return fail(UGS_E_BAD_ARG, "bad arguments to sample_batch");
    DeviceCtx dc;
    if (int rc = device_ctx(dc)) return rc;
    const int64_t G = num_graphs, rows = G * (int64_t)m_per_graph, cap = edge_capacity;
    if (rows > 0 && (!nodes || (cap > 0 && (!edge_index_out || !edge_src_global)))) return fail(UGS_E_BAD_ARG, "null output pointer");
    // The real lookup reads every column (content hash: ~3 ms for 20 M columns).  A batch whose sampled words match a remembered one
    // starts on that batch's plan at once, the lookup runs beside it on a helper thread, and the results stand only if the lookup
    // returns the same plan; otherwise the streams are drained and the call runs again on the right one.
    ugs_plan *plan = nullptr, *guess = nullptr;
    if (rows > 0 && !debug_on() && std::getenv("UGS_NO_SPECULATION") == nullptr && std::getenv("UGS_NO_BATCH_INDEX") == nullptr)
        guess = peek_batch_plan(edge_index, edge_index + row_stride, num_cols, ptr, G, k, dc.id);
    std::thread lookup_thread;
    int lookup_rc = UGS_OK;
    std::string lookup_err;
    if (guess) {
        const int dev_tl = t_device; const hipStream_t st_tl = t_job_stream; const bool set_tl = t_job_stream_set;
        try {
            lookup_thread = std::thread([&, dev_tl, st_tl, set_tl] {
                t_device = dev_tl >= 0 ? dev_tl : dc.id; t_job_stream = st_tl; t_job_stream_set = set_tl;
                lookup_rc = ugs_plan_create_batch(edge_index, row_stride, num_cols, ptr, num_graphs, k, &plan);
                if (lookup_rc != UGS_OK) lookup_err = t_err;
            });
        } catch (...) { plan_unref(guess); guess = nullptr; }          // no thread to be had: look up first, as without a guess
    }
    if (!guess) if (int rc = ugs_plan_create_batch(edge_index, row_stride, num_cols, ptr, num_graphs, k, &plan)) return rc;
    StreamCall sc;
    sc.dc = dc; sc.G = G; sc.m = m_per_graph; sc.k = k; sc.mode = mode; sc.extra = 0; sc.seed = seed; sc.cap = cap;
    sc.nodes = nodes; sc.edge_index = edge_index_out; sc.edge_ptr = edge_ptr; sc.sample_ptr = sample_ptr; sc.edge_src = edge_src_global;
    sc.total_out = total_edges_out;
    hipStream_t s = dc.stream;
    auto body = [&](ugs_plan *p) { return stream_rows(p, sc); };
    auto drain = [&] { if (sc.cs) (void)hipStreamSynchronize(sc.cs); (void)hipStreamSynchronize(s); };
    int rc = UGS_OK;
    if (guess) {
        std::string guess_err;
        rc = body(guess);
        if (rc != UGS_OK) guess_err = t_err;
        lookup_thread.join();
        if (lookup_rc != UGS_OK) { drain(); rc = fail(lookup_rc, lookup_err); }             // the call's own error (what the two-phase call reports)
        else if (plan != guess) { drain(); g_spec_wrong.fetch_add(1); rc = body(plan); }      // a different batch after all: once more, on its plan
        else if (rc != UGS_OK) rc = fail(rc, guess_err);
        else g_spec_kept.fetch_add(1);
        plan_unref(guess);
    } else rc = body(plan);
    if (rc != UGS_OK) drain();                                      // nothing may still read the pool buffers
    sc.release();
    if (plan) plan_unref(plan);
    return rc;
}

/* sample() of the handle API (reference src/sampler.cpp:91-290) the same way: rows in chunks, copy-out beside the walks. */
int ugs_sample_stream(int64_t handle, int m_per_graph, int k, int edge_mode, int64_t base_offset, int seed, int64_t edge_capacity,
                      int64_t *nodes, int64_t *edge_index_out, int64_t *edge_ptr, int64_t *edge_src, int64_t *total_edges_out) {
    if (edge_mode < 0 || edge_mode > 2) return fail(UGS_E_BAD_MODE, "edge_mode must be one of: 'local', 'flat', 'global'");
    if (m_per_graph < 0) return fail(UGS_E_BAD_ARG, "m_per_graph must be >= 0");
    if (edge_capacity < 0 || !edge_ptr || !total_edges_out) return fail(UGS_E_BAD_ARG, "null output pointer or negative edge_capacity");
    if (m_per_graph > 0 && (!nodes || (edge_capacity > 0 && (!edge_index_out || !edge_src)))) return fail(UGS_E_BAD_ARG, "null output pointer");
    auto g = lookup(handle);
    if (!g) return fail(UGS_E_INVALID_HANDLE, "Invalid preproc handle");
    if (g->n == 0) return fail(UGS_E_NO_ROOTS, "No viable roots available");
    if (k < 1) return fail(UGS_E_BAD_ARG, "k must be >= 1");
    if (k > UGS_KMAX) return fail(UGS_E_UNSUPPORTED, "k > 32 is not supported by the HIP sampler");
    ugs_plan *plan = nullptr;
    if (int rc = ugs_plan_create_handle(handle, &plan)) return rc;
    StreamCall sc;
    if (int rc = device_ctx(sc.dc)) { plan_unref(plan); return rc; }
    sc.G = plan->G; sc.m = m_per_graph; sc.k = k; sc.mode = edge_mode; sc.extra = edge_mode == UGS_EDGE_GLOBAL ? base_offset : 0; sc.seed = seed;
    sc.cap = edge_capacity;
    sc.nodes = nodes; sc.edge_index = edge_index_out; sc.edge_ptr = edge_ptr; sc.sample_ptr = nullptr; sc.edge_src = edge_src;
    sc.total_out = total_edges_out;
    const int rc = stream_rows(plan, sc);
    if (rc != UGS_OK) { if (sc.cs) (void)hipStreamSynchronize(sc.cs); (void)hipStreamSynchronize(sc.dc.stream); }
    sc.release();
    plan_unref(plan);
    return rc;
}

// ---- epsilon_uniform_sampler.sample_batch (reference src/samplers/epsilon_uniform_sampler/src/epsilon_uniform_sampler.cpp) ----
int ugs_eps_sample_batch_begin(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr, int64_t num_graphs,
                               int m_per_graph, int k, int mode, uint64_t seed, double epsilon, ugs_job **job_out, int64_t *total_edges_out) {
    if (!job_out || !ptr || num_graphs < 0 || num_cols < 0 || (num_cols > 0 && !edge_index)) return fail(UGS_E_BAD_ARG, "bad arguments to sample_batch");
    if (!(epsilon > 0.0 && epsilon <= 1.0)) return fail(UGS_E_BAD_ARG, "epsilon must be in (0, 1]");
    if (m_per_graph < 0) return fail(UGS_E_BAD_ARG, "m_per_graph must be >= 0");
    if (k < 1) return fail(UGS_E_BAD_ARG, "k must be >= 1");
    if (k > UGS_KMAX) return fail(UGS_E_UNSUPPORTED, "k > 32 is not supported by the HIP sampler");
    if (num_cols >= ((int64_t)1 << 30)) return fail(UGS_E_UNSUPPORTED, "batch too large: columns must be < 2^30");
    DeviceCtx dc;
    if (int rc = device_ctx(dc)) return rc;
    const int64_t G = num_graphs, E = num_cols;
    const int64_t *src = edge_index, *dst = edge_index + row_stride;
    std::vector<int64_t> cstart, cols_of;
    assign_columns(src, dst, E, ptr, G, cstart, cols_of);
    // adjacency lists in the reference's push_back order (:152-176): column order, source row then destination row
    int64_t nrows = 0, nnz = 2 * (int64_t)cols_of.size();
    for (int64_t g = 0; g < G; ++g) nrows += std::max<int64_t>(ptr[g + 1] - ptr[g], 0) + 1;
    if (nnz >= (int64_t)INT32_MAX) return fail(UGS_E_UNSUPPORTED, "batch too large");
    size_t off_desc = 0;
    size_t off_row = align_up(off_desc + (size_t)std::max<int64_t>(G, 1) * sizeof(UgsGraphDesc));
    size_t off_nbr = align_up(off_row + (size_t)std::max<int64_t>(nrows, 1) * sizeof(int64_t));
    size_t off_ecs = align_up(off_nbr + (size_t)std::max<int64_t>(nnz, 1) * sizeof(int32_t));
    size_t total = align_up(off_ecs + (size_t)std::max<int64_t>(nnz, 1) * sizeof(int32_t));
    std::vector<char> host(total, 0);
    auto *desc = reinterpret_cast<UgsGraphDesc *>(host.data() + off_desc);
    auto *rowp = reinterpret_cast<int64_t *>(host.data() + off_row);
    auto *nbr = reinterpret_cast<int32_t *>(host.data() + off_nbr);
    auto *ecs = reinterpret_cast<int32_t *>(host.data() + off_ecs);
    int64_t rb = 0, ab = 0;
    std::vector<int64_t> wr;
    for (int64_t g = 0; g < G; ++g) {
        const int64_t lo = ptr[g], n = std::max<int64_t>(ptr[g + 1] - ptr[g], 0);
        if (n >= (int64_t)INT32_MAX) return fail(UGS_E_UNSUPPORTED, "graph too large");
        UgsGraphDesc &d = desc[g];
        d.node_lo = lo; d.rbase = rb; d.vbase = 0; d.viable_base = 0; d.n = (int32_t)n; d.level = 0; d.n_viable = 0; d.pad = 0;
        const int64_t c0 = cstart[(size_t)g], cn = cstart[(size_t)g + 1] - c0;
        for (int64_t r = 0; r <= n; ++r) rowp[rb + r] = 0;
        for (int64_t t = 0; t < cn; ++t) { const int64_t j = cols_of[(size_t)(c0 + t)]; ++rowp[rb + (src[j] - lo) + 1]; ++rowp[rb + (dst[j] - lo) + 1]; }
        for (int64_t r = 0; r < n; ++r) rowp[rb + r + 1] += rowp[rb + r];
        wr.assign(rowp + rb, rowp + rb + n);
        for (int64_t t = 0; t < cn; ++t) {
            const int64_t j = cols_of[(size_t)(c0 + t)], u = src[j] - lo, v = dst[j] - lo;
            int64_t a = ab + wr[(size_t)u]++; nbr[a] = (int32_t)v; ecs[a] = (int32_t)(2 * j);
            int64_t b = ab + wr[(size_t)v]++; nbr[b] = (int32_t)u; ecs[b] = (int32_t)(2 * j + 1);
        }
        for (int64_t r = 0; r <= n; ++r) rowp[rb + r] += ab;
        rb += n + 1; ab += 2 * cn;
    }
    auto *j = new ugs_job();
    j->dc = dc; j->eps = true; j->batch = true; j->m = m_per_graph; j->k = k; j->mode = mode; j->G = G;
    j->rows = G * (int64_t)m_per_graph;
    auto bail = [&](int rc) { free_job(j); return rc; };
    if (int rc = pool_get(total, dc.id, j->eps_blob)) return bail(rc);
    hipError_t e = hipMemcpy(j->eps_blob.p, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(fail_hip(e, "hipMemcpy"));
    char *base = static_cast<char *>(j->eps_blob.p);
    if (int rc = pool_get((size_t)(j->rows * k + j->rows + 1) * sizeof(int64_t), dc.id, j->nodes)) return bail(rc);
    j->d_eptr = static_cast<int64_t *>(j->nodes.p) + j->rows * k;
    if (int rc = pool_get((size_t)std::max<int64_t>(j->rows, 1) * sizeof(uint32_t), dc.id, j->eps_counts)) return bail(rc);
    if (int rc = pool_get((size_t)ugs_scan_tmp_words(j->rows) * sizeof(int64_t), dc.id, j->eps_scantmp)) return bail(rc);
    UgsEpsLaunch &l = j->eps_l;
    l.graphs = reinterpret_cast<const UgsGraphDesc *>(base + off_desc);
    l.rowptr = reinterpret_cast<const int64_t *>(base + off_row);
    l.nbr = reinterpret_cast<const int32_t *>(base + off_nbr);
    l.ecs = reinterpret_cast<const int32_t *>(base + off_ecs);
    l.num_graphs = G; l.m = m_per_graph; l.k = k; l.mode = mode;
    l.max_attempts = std::max(10, (int)(10.0 / epsilon));
    l.seed = seed; l.epsilon = epsilon; l.rows = j->rows;
    l.nodes = static_cast<int64_t *>(j->nodes.p); l.counts = static_cast<uint32_t *>(j->eps_counts.p);
    l.edge_ptr = nullptr; l.edge_index = nullptr; l.edge_src = nullptr; l.ld = 0;
    e = ugs_eps_launch(l, 0, dc.cus, dc.stream);
    if (e != hipSuccess) return bail(fail_hip(e, "ugs_eps_walk"));
    e = ugs_launch_scan(l.counts, j->rows, j->d_eptr, static_cast<int64_t *>(j->eps_scantmp.p), dc.stream);
    if (e != hipSuccess) return bail(fail_hip(e, "ugs_launch_scan"));
    e = hipMemcpyAsync(&j->total, j->d_eptr + j->rows, sizeof(int64_t), hipMemcpyDeviceToHost, dc.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dc.stream);
    if (e != hipSuccess) return bail(fail_hip(e, "epsilon walk"));
    *job_out = j;
    if (total_edges_out) *total_edges_out = j->total;
    return UGS_OK;
}

int ugs_eps_sample_batch_finish(ugs_job *job, int64_t *nodes, int64_t *edge_index, int64_t *edge_ptr, int64_t *sample_ptr,
                                int64_t *edge_src, int dst_is_device) {
    if (!job || !job->eps) return fail(UGS_E_BAD_ARG, "not an epsilon job");
    return finish_common(job, nodes, edge_index, edge_ptr, sample_ptr, edge_src, dst_is_device);
}

}  // extern "C"
