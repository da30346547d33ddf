// ugs_eps.hip -- gfx950 kernels of the epsilon-uniform connected-subgraph sampler.
//
// Contract: the reference's epsilon_uniform_sampler (src/samplers/epsilon_uniform_sampler/src/epsilon_uniform_sampler.cpp):
// frontier growth sample_connected_subgraph_rw (:18-87), acceptance min(1, eps/(w+eps)) (:238), max(10, 10/eps) attempts
// (:207), sorted nodes (:256), edges = batch columns inside the sample in column order (:265-291).
//
// MI355X mapping: samples are independent and tiny (k <= 32 vertices, a handful of short adjacency rows), so ONE LANE owns
// one sample (64 samples per wavefront); per-sample vertex and frontier lists live in LDS in a [slot][lane] layout
// (conflict-free); random numbers come from a counter-based generator keyed by (seed, row, attempt), so a row never
// depends on which lane, wave or GPU computes it.  Edge extraction is a count pass (fused into the walk), a scan, and a
// fill pass that writes each sample's edges and orders them by column inside the sample's own segment.
#include "ugs_device.h"

namespace {

constexpr int EPS_BLOCK = 128;

// splitmix64 finaliser as the keyed counter -> stream seed; xorshift64* as the stream
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
struct CRng {
    uint64_t s;
    __device__ __forceinline__ void init(uint64_t seed, uint64_t row, uint64_t attempt) {
        s = mix64(mix64(seed ^ mix64(row)) + attempt);
        if (s == 0) s = 0x9e3779b97f4a7c15ull;
    }
    __device__ __forceinline__ uint64_t next() {
        uint64_t x = s;
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        s = x;
        return x * 2685821657736338717ull;
    }
    // unbiased integer in [0, n), n >= 1 (multiply-shift with rejection)
    __device__ __forceinline__ uint32_t below(uint32_t n) {
        uint64_t m = (uint64_t)(uint32_t)(next() >> 32) * (uint64_t)n;
        uint32_t lo = (uint32_t)m;
        if (lo < n) {
            const uint32_t t = (0u - n) % n;
            while (lo < t) { m = (uint64_t)(uint32_t)(next() >> 32) * (uint64_t)n; lo = (uint32_t)m; }
        }
        return (uint32_t)(m >> 32);
    }
    __device__ __forceinline__ double unit() { return (double)(next() >> 11) * 0x1p-53; }   // [0, 1)
};

struct EpsArgs {
    const UgsGraphDesc *graphs;   // node_lo, rbase, n
    const int64_t *rowptr;
    const int32_t *nbr;           // neighbour (local id), rows in the reference's push_back order
    const int32_t *ecs;           // 2 * batch column + side (0: this row is the column's source endpoint)
    int64_t num_graphs;
    int32_t m, k, mode, max_attempts;
    uint64_t seed;
    double epsilon;
    int64_t rows;
    int64_t *nodes;               // [rows, k] batch node ids, ascending; -1 rows for failed samples
    uint32_t *counts;             // [rows]
    const int64_t *edge_ptr;      // fill pass
    int64_t *edge_index, *edge_src;
    int64_t ld;
};

__global__ __launch_bounds__(EPS_BLOCK) void ugs_eps_walk(EpsArgs a) {
    __shared__ int32_t s_nodes[UGS_KMAX * EPS_BLOCK];
    __shared__ int32_t s_front[UGS_KMAX * EPS_BLOCK];
    const int tid = (int)threadIdx.x;
    int32_t *nd = s_nodes + tid, *fr = s_front + tid;      // element j at [j * EPS_BLOCK]
    const int k = a.k;
    for (int64_t row = (int64_t)blockIdx.x * EPS_BLOCK + tid; row < a.rows; row += (int64_t)gridDim.x * EPS_BLOCK) {
        const int64_t gi = row / a.m;
        const UgsGraphDesc gd = a.graphs[gi];
        const int n = gd.n;
        bool success = false;
        if (n >= k) {
            for (int attempt = 0; attempt < a.max_attempts && !success; ++attempt) {
                CRng rng;
                rng.init(a.seed, (uint64_t)row, (uint64_t)attempt);
                int size = 1, fsz = 1;
                const int start = (int)rng.below((uint32_t)n);
                nd[0] = start; fr[0] = start;
                double weight = 1.0;
                weight *= (1.0 / n);
                int tries = 0;
                bool dead = false;
                while (size < k && tries < k * 100) {
                    ++tries;
                    if (fsz == 0) { dead = true; break; }
                    const int fi = (int)rng.below((uint32_t)fsz);
                    const int u = fr[fi * EPS_BLOCK];
                    const int64_t r0 = a.rowptr[gd.rbase + u], r1 = a.rowptr[gd.rbase + u + 1];
                    int cnt = 0;
                    for (int64_t p = r0; p < r1; ++p) {
                        const int v = a.nbr[p];
                        bool in = false;
                        for (int j = 0; j < size; ++j) in = in || nd[j * EPS_BLOCK] == v;
                        cnt += in ? 0 : 1;
                    }
                    if (cnt == 0) {                                    // exhausted frontier vertex: drop it, keep the order
                        for (int j = fi; j + 1 < fsz; ++j) fr[j * EPS_BLOCK] = fr[(j + 1) * EPS_BLOCK];
                        --fsz;
                        continue;
                    }
                    int pick = (int)rng.below((uint32_t)cnt), chosen = -1;
                    for (int64_t p = r0; p < r1; ++p) {
                        const int v = a.nbr[p];
                        bool in = false;
                        for (int j = 0; j < size; ++j) in = in || nd[j * EPS_BLOCK] == v;
                        if (!in) { if (pick == 0) { chosen = v; break; } --pick; }
                    }
                    nd[size * EPS_BLOCK] = chosen; ++size;
                    fr[fsz * EPS_BLOCK] = chosen; ++fsz;
                    weight *= (1.0 / fsz) * (1.0 / cnt);
                }
                if (dead || size < k) continue;
                const double acc = fmin(1.0, a.epsilon / (weight + a.epsilon));
                if (rng.unit() <= acc) success = true;
            }
        }
        int64_t *out = a.nodes + row * k;
        uint32_t ecount = 0;
        if (success) {
            for (int i = 1; i < k; ++i) {                               // ascending order (insertion sort)
                const int x = nd[i * EPS_BLOCK];
                int j = i - 1;
                while (j >= 0 && nd[j * EPS_BLOCK] > x) { nd[(j + 1) * EPS_BLOCK] = nd[j * EPS_BLOCK]; --j; }
                nd[(j + 1) * EPS_BLOCK] = x;
            }
            for (int j = 0; j < k; ++j) {
                const int u = nd[j * EPS_BLOCK];
                out[j] = gd.node_lo + u;
                const int64_t r0 = a.rowptr[gd.rbase + u], r1 = a.rowptr[gd.rbase + u + 1];
                for (int64_t p = r0; p < r1; ++p) {
                    if (a.ecs[p] & 1) continue;                          // count every column once, at its source endpoint
                    const int v = a.nbr[p];
                    bool in = false;
                    for (int t = 0; t < k; ++t) in = in || nd[t * EPS_BLOCK] == v;
                    ecount += in ? 1u : 0u;
                }
            }
        } else {
            for (int j = 0; j < k; ++j) out[j] = -1;
        }
        a.counts[row] = ecount;
    }
}

__global__ __launch_bounds__(EPS_BLOCK) void ugs_eps_fill(EpsArgs a) {
    const int k = a.k;
    for (int64_t row = (int64_t)blockIdx.x * EPS_BLOCK + threadIdx.x; row < a.rows; row += (int64_t)gridDim.x * EPS_BLOCK) {
        const int64_t e0 = a.edge_ptr[row], e1 = a.edge_ptr[row + 1];
        if (e1 == e0) continue;
        const int64_t gi = row / a.m;
        const UgsGraphDesc gd = a.graphs[gi];
        const int64_t *nrow = a.nodes + row * k;
        int64_t w = e0;
        for (int j = 0; j < k; ++j) {
            const int64_t ug = nrow[j];
            const int u = (int)(ug - gd.node_lo);
            const int64_t r0 = a.rowptr[gd.rbase + u], r1 = a.rowptr[gd.rbase + u + 1];
            for (int64_t p = r0; p < r1; ++p) {
                const int32_t cs = a.ecs[p];
                if (cs & 1) continue;
                const int64_t vg = gd.node_lo + a.nbr[p];
                int l = -1;
                for (int t = 0; t < k; ++t) if (nrow[t] == vg) { l = t; break; }
                if (l < 0) continue;
                a.edge_index[w] = a.mode == 0 ? (int64_t)j : ug;
                a.edge_index[a.ld + w] = a.mode == 0 ? (int64_t)l : vg;
                a.edge_src[w] = (int64_t)(cs >> 1);
                ++w;
            }
        }
        // column order inside the sample's own segment (insertion sort; segments hold a handful of edges)
        for (int64_t i = e0 + 1; i < e1; ++i) {
            const int64_t s = a.edge_src[i], x = a.edge_index[i], y = a.edge_index[a.ld + i];
            int64_t j = i - 1;
            while (j >= e0 && a.edge_src[j] > s) {
                a.edge_src[j + 1] = a.edge_src[j]; a.edge_index[j + 1] = a.edge_index[j]; a.edge_index[a.ld + j + 1] = a.edge_index[a.ld + j];
                --j;
            }
            a.edge_src[j + 1] = s; a.edge_index[j + 1] = x; a.edge_index[a.ld + j + 1] = y;
        }
    }
}

}  // namespace

struct UgsEpsLaunch {
    const UgsGraphDesc *graphs; const int64_t *rowptr; const int32_t *nbr; const int32_t *ecs; int64_t num_graphs;
    int32_t m, k, mode, max_attempts; uint64_t seed; double epsilon; int64_t rows;
    int64_t *nodes; uint32_t *counts; const int64_t *edge_ptr; int64_t *edge_index; int64_t *edge_src; int64_t ld;
};

hipError_t ugs_eps_launch(const UgsEpsLaunch &l, int fill, int cus, hipStream_t s) {
    if (l.rows <= 0) return hipSuccess;
    EpsArgs a;
    a.graphs = l.graphs; a.rowptr = l.rowptr; a.nbr = l.nbr; a.ecs = l.ecs; a.num_graphs = l.num_graphs;
    a.m = l.m; a.k = l.k; a.mode = l.mode; a.max_attempts = l.max_attempts; a.seed = l.seed; a.epsilon = l.epsilon; a.rows = l.rows;
    a.nodes = l.nodes; a.counts = l.counts; a.edge_ptr = l.edge_ptr; a.edge_index = l.edge_index; a.edge_src = l.edge_src; a.ld = l.ld;
    int64_t grid = (l.rows + EPS_BLOCK - 1) / EPS_BLOCK;
    const int64_t cap = (int64_t)(cus > 0 ? cus : 256) * 8;
    if (grid > cap) grid = cap;
    if (fill) hipLaunchKernelGGL(ugs_eps_fill, dim3((unsigned)grid), dim3(EPS_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(ugs_eps_walk, dim3((unsigned)grid), dim3(EPS_BLOCK), 0, s, a);
    return hipGetLastError();
}
