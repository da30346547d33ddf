// ugs_batch.hip -- the input side of sample_batch for batches of SMALL graphs, on the device.
//
// Replaces, for every call whose batch was not seen before, the host's per-graph slicing, renumbering, hashing and CSR
// construction (reference src/ugs_sampler_batch_extension.cpp:41-75 slice_and_renumber_edge_index_with_map, include/cache.hpp:81-109
// hash_graph, src/preproc.cpp:32-86 build_csr, :88-140 the degree ordering) by one pass over edge_index + ptr:
//
//   ugs_bp_assign   one thread per column: the graph that owns it -- both endpoints inside [ptr[g], ptr[g+1]); with a
//                   non-decreasing ptr the node ranges are disjoint, so a binary search finds the only candidate -- plus per
//                   graph the number of its columns and the first / last column index (atomics)
//   ugs_bp_offsets  one block: exclusive prefix of the column counts (a graph's CSR entries start at twice that)
//   ugs_bp_build    one 256-thread block per graph: its columns in column order (ordered compaction of [first, last]), renumbered
//                   to local ids in LDS; the FNV-1a key of the reference's LRU over (n, #columns, renumbered columns) -- inherently
//                   sequential, one lane, 2 multiplies per column; degrees, row pointer, the rank of every vertex in the reference's
//                   ordering (ascending (degree, id): ugs_host.cpp order_by_degree) and the CSR of the symmetrised multigraph with
//                   entries in COLUMN order (u's row then v's row per column, both entries of a self loop) written straight into
//                   the plan's arrays: rowptr (absolute), adj = (w, rank(w)), adjf = (w, batch column)
//
// The host then sees G keys (16 bytes per graph come back), replays the LRU on them and points each graph's descriptor at the
// root records its cached preprocessing left in HBM (ugs_host.cpp: device_batch_plan).  A key covers the whole content of a
// graph of at most 1000 columns (cache.hpp:100 samples longer ones), so a cached graph with the same key HAS this CSR; graphs
// beyond the limits below send the whole batch down the host path.
#include "ugs_device.h"

namespace {

constexpr int kBpBlock = 256;

__global__ __launch_bounds__(256) void ugs_bp_assign(const int64_t *src, const int64_t *dst, int64_t E, const int64_t *ptr, int64_t G,
                                                     int32_t *owner, uint32_t *cnt, uint32_t *jmin, uint32_t *jmax) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= E) return;
    const int64_t u = src[j], v = dst[j];
    int32_t g = -1;
    if (u >= ptr[0] && u < ptr[G]) {
        int64_t lo = 0, hi = G;                               // last g with ptr[g] <= u (ptr is non-decreasing)
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (ptr[mid] <= u) lo = mid; else hi = mid; }
        // empty graphs share their ptr value with the next one: the search ends on the LAST graph starting at or below u,
        // which is the one whose range [ptr[g], ptr[g+1]) can hold u
        if (u >= ptr[lo] && u < ptr[lo + 1] && v >= ptr[lo] && v < ptr[lo + 1]) g = (int32_t)lo;
    }
    owner[j] = g;
    if (g >= 0) {
        atomicAdd(&cnt[g], 1u);
        atomicMin(&jmin[g], (uint32_t)j);
        atomicMax(&jmax[g], (uint32_t)j);
    }
}

__global__ __launch_bounds__(1024) void ugs_bp_offsets(const uint32_t *cnt, int64_t G, uint32_t *cstart /* [G+1] */) {
    __shared__ uint32_t sh[1024 / 64];
    __shared__ uint32_t carry_sh;
    if (threadIdx.x == 0) carry_sh = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t base = 0; base < G; base += 1024) {
        const int64_t g = base + threadIdx.x;
        const uint32_t x = g < G ? cnt[g] : 0u;
        uint32_t incl = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
        if (lane == 63) sh[wv] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int i = 0; i < 1024 / 64; ++i) { if (i < wv) woff += sh[i]; tot += sh[i]; }
        const uint32_t carry = carry_sh;
        if (g < G) cstart[g] = carry + woff + incl - x;
        __syncthreads();
        if (threadIdx.x == 0) carry_sh = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) cstart[G] = carry_sh;
}

struct BpBuild {
    const int64_t *src, *dst, *ptr;
    const int32_t *owner;
    const uint32_t *cnt, *jmin, *jmax, *cstart;
    const int64_t *rstart;       // [G] first rowptr entry of graph g (host: prefix of n_g + 1 over the non-degenerate graphs)
    int32_t k;
    int64_t *rowptr;
    int2 *adj, *adjf;
    unsigned long long *keys;    // [G] FNV-1a key (0 for degenerate graphs: never looked up)
    uint32_t *maxdeg;            // [G]
    uint32_t *flag;              // set when a graph exceeds the limits of this path
};

constexpr int kBpMaxCols = UGS_BATCH_PASS_MAX_COLS, kBpMaxN = UGS_BATCH_PASS_MAX_N;

__global__ __launch_bounds__(kBpBlock) void ugs_bp_build(BpBuild a) {
    __shared__ uint16_t LU[kBpMaxCols], LV[kBpMaxCols];
    __shared__ int32_t LC[kBpMaxCols];                       // batch column of local column t
    __shared__ uint16_t DEG[kBpMaxN], RNK[kBpMaxN];
    __shared__ uint32_t RP[kBpMaxN + 1];
    __shared__ uint32_t wsum[kBpBlock / 64];
    __shared__ uint32_t run_sh;
    const int g = (int)blockIdx.x;
    const int64_t lo = a.ptr[g], n64 = a.ptr[g + 1] - lo;
    if (n64 <= 0 || n64 < a.k) { if (threadIdx.x == 0) { a.keys[g] = 0ull; a.maxdeg[g] = 0u; } return; }   // degenerate: rows of -1, nothing to build
    const uint32_t cn = a.cnt[g];
    if (n64 > kBpMaxN || cn > (uint32_t)kBpMaxCols) { if (threadIdx.x == 0) atomicOr(a.flag, 1u); return; }
    const int n = (int)n64;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // 1. the graph's columns in column order: ordered compaction of the span [first, last] of its columns
    if (tid == 0) run_sh = 0u;
    for (int x = tid; x < n; x += kBpBlock) DEG[x] = 0;
    __syncthreads();
    if (cn) {
        const uint32_t j0 = a.jmin[g], j1 = a.jmax[g];
        for (uint32_t base = j0; base <= j1; base += kBpBlock) {
            const uint32_t j = base + (uint32_t)tid;
            const bool mine = j <= j1 && a.owner[j] == g;
            const uint64_t mk = __ballot(mine);
            if (lane == 0) wsum[wv] = (uint32_t)__popcll(mk);
            __syncthreads();
            uint32_t off = run_sh;
            for (int i = 0; i < wv; ++i) off += wsum[i];
            if (mine) {
                const uint32_t t = off + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
                LU[t] = (uint16_t)(a.src[j] - lo);
                LV[t] = (uint16_t)(a.dst[j] - lo);
                LC[t] = (int32_t)j;
            }
            __syncthreads();
            if (tid == 0) { uint32_t tot = 0; for (int i = 0; i < kBpBlock / 64; ++i) tot += wsum[i]; run_sh += tot; }
            __syncthreads();
        }
    }
    // 2. the LRU key (reference include/cache.hpp:81-109; all columns: cn <= 1000 on this path): one lane, while the others count degrees
    if (tid == 0) {
        const unsigned long long prime = 1099511628211ull;
        unsigned long long h = 14695981039346656037ull;
        h = (h ^ (unsigned long long)n) * prime;
        h = (h ^ (unsigned long long)cn) * prime;
        for (uint32_t t = 0; t < cn; ++t) { h = (h ^ (unsigned long long)LU[t]) * prime; h = (h ^ (unsigned long long)LV[t]) * prime; }
        a.keys[g] = h;
    }
    // 3. degrees of the symmetrised multigraph (a self loop adds two entries to its row: reference src/preproc.cpp:47-60).  16-bit
    //    counters packed two to a word would need word atomics; one lane per vertex counting its own row is order-free and tiny
    for (int x = tid; x < n; x += kBpBlock) {
        uint32_t d = 0;
        for (uint32_t t = 0; t < cn; ++t) d += (LU[t] == x ? 1u : 0u) + (LV[t] == x ? 1u : 0u);
        DEG[x] = (uint16_t)d;
    }
    __syncthreads();
    // 4. row pointer (exclusive scan of the degrees) and the maximum degree
    {
        uint32_t carry = 0, mx = 0;
        for (int base = 0; base < n; base += kBpBlock) {
            const int x = base + tid;
            const uint32_t d = x < n ? DEG[x] : 0u;
            mx = d > mx ? d : mx;
            uint32_t incl = d;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) { const uint32_t y = __shfl_up(incl, s, 64); if (lane >= s) incl += y; }
            if (lane == 63) wsum[wv] = incl;
            __syncthreads();
            uint32_t woff = 0, tot = 0;
            for (int i = 0; i < kBpBlock / 64; ++i) { if (i < wv) woff += wsum[i]; tot += wsum[i]; }
            if (x < n) RP[x] = carry + woff + incl - d;
            carry += tot;
            __syncthreads();
        }
        if (tid == 0) RP[n] = carry;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { const uint32_t y = __shfl_xor(mx, s, 64); mx = y > mx ? y : mx; }
        if (lane == 0) wsum[wv] = mx;
        __syncthreads();
        if (tid == 0) { uint32_t m = 0; for (int i = 0; i < kBpBlock / 64; ++i) m = wsum[i] > m ? wsum[i] : m; a.maxdeg[g] = m; }
    }
    // 5. rank(x) in the reference's vertex order = ascending (CSR degree, vertex id) (src/preproc.cpp:88-140 restated: ugs_host.cpp)
    for (int x = tid; x < n; x += kBpBlock) {
        const uint32_t dx = DEG[x];
        uint32_t r = 0;
        for (int y = 0; y < n; ++y) { const uint32_t dy = DEG[y]; r += (dy < dx || (dy == dx && y < x)) ? 1u : 0u; }
        RNK[x] = (uint16_t)r;
    }
    __syncthreads();
    // 6. the plan's arrays: rows in column order (per column u's row first, then v's: both entries of a self loop land in its row)
    const int64_t abase = 2ll * (int64_t)a.cstart[g], rbase = a.rstart[g];
    for (int x = tid; x <= n; x += kBpBlock) a.rowptr[rbase + x] = abase + (int64_t)RP[x];
    for (int x = tid; x < n; x += kBpBlock) {
        int64_t p = abase + (int64_t)RP[x];
        for (uint32_t t = 0; t < cn; ++t) {
            const int u = LU[t], v = LV[t];
            if (u == x) { a.adj[p] = make_int2(v, (int)RNK[v]); a.adjf[p] = make_int2(v, LC[t]); ++p; }
            if (v == x) { a.adj[p] = make_int2(u, (int)RNK[u]); a.adjf[p] = make_int2(u, LC[t]); ++p; }
        }
    }
}

}  // namespace

hipError_t ugs_launch_batch_pass(const int64_t *d_src, const int64_t *d_dst, int64_t E, const int64_t *d_ptr, int64_t G, int k,
                                 int32_t *d_owner, uint32_t *d_cnt, uint32_t *d_jmin, uint32_t *d_jmax, uint32_t *d_cstart,
                                 const int64_t *d_rstart, int64_t *d_rowptr, int2 *d_adj, int2 *d_adjf, unsigned long long *d_keys,
                                 uint32_t *d_maxdeg, uint32_t *d_flag, hipStream_t s) {
    if (G <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_cnt, 0, (size_t)G * sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_jmin, 0xFF, (size_t)G * sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_jmax, 0, (size_t)G * sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (E > 0) hipLaunchKernelGGL(ugs_bp_assign, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, d_src, d_dst, E, d_ptr, G, d_owner, d_cnt, d_jmin, d_jmax);
    hipLaunchKernelGGL(ugs_bp_offsets, dim3(1), dim3(1024), 0, s, (const uint32_t *)d_cnt, G, d_cstart);
    BpBuild a{};
    a.src = d_src; a.dst = d_dst; a.ptr = d_ptr; a.owner = d_owner; a.cnt = d_cnt; a.jmin = d_jmin; a.jmax = d_jmax; a.cstart = d_cstart;
    a.rstart = d_rstart; a.k = k; a.rowptr = d_rowptr; a.adj = d_adj; a.adjf = d_adjf; a.keys = d_keys; a.maxdeg = d_maxdeg; a.flag = d_flag;
    hipLaunchKernelGGL(ugs_bp_build, dim3((unsigned)G), dim3(kBpBlock), 0, s, a);
    return hipGetLastError();
}
