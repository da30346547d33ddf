// ugs_batch.hip -- the input side of sample_batch for batches of SMALL graphs, on the device.
//
// Replaces, for every call whose batch was not seen before, the host's per-graph slicing, renumbering, hashing and CSR
// construction (reference src/ugs_sampler_batch_extension.cpp:41-75 slice_and_renumber_edge_index_with_map, include/cache.hpp:81-109
// hash_graph, src/preproc.cpp:32-86 build_csr, :88-140 the degree ordering) by one pass over edge_index + ptr:
//
//   ugs_bp_assign   one thread per column: the graph that owns it -- both endpoints inside [ptr[g], ptr[g+1]); with a
//                   non-decreasing ptr the node ranges are disjoint, so a binary search finds the only candidate -- plus per
//                   graph the number of its columns and the first / last column index (atomics)
//                   (only for large batches; for G * E up to a few million the build kernel finds its columns itself, as the
//                   reference's slicing does, and this launch, its memset and its atomics are saved)
//   ugs_bp_build    one 256-thread block per graph: its columns in column order (ordered compaction of the span [first, last] of
//                   its columns, or of all columns), renumbered
//                   to local ids in LDS; the FNV-1a key of the reference's LRU over (n, #columns, renumbered columns) -- inherently
//                   sequential: 2 multiplies per column on the scalar unit; degrees, row pointer, the rank of every vertex in the reference's
//                   ordering (ascending (degree, id): ugs_host.cpp order_by_degree) and the CSR of the symmetrised multigraph with
//                   entries in COLUMN order (u's row then v's row per column, both entries of a self loop) written straight into
//                   the plan's arrays: rowptr (absolute), adj = (w, rank(w)), adjf = (w, batch column)
//
// The host then sees G keys (16 bytes per graph come back), replays the LRU on them and points each graph's descriptor at the
// root records its cached preprocessing left in HBM (ugs_host.cpp: device_batch_plan).  A key covers the whole content of a
// graph of at most 1000 columns (cache.hpp:100 samples longer ones), so a cached graph with the same key HAS this CSR; graphs
// beyond the limits below send the whole batch down the host path.
#include "ugs_device.h"
#include <cstdlib>

namespace {

constexpr int kBpBlock = 256;

__global__ __launch_bounds__(256) void ugs_bp_assign(const int64_t *src, const int64_t *dst, int64_t E, const int64_t *ptr, int64_t G,
                                                     int32_t *owner, uint32_t *cnt, uint32_t *jminc /* 0xFFFFFFFF - first column: zero-initialised like the rest */,
                                                     uint32_t *jmax) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = j < E;
    const int64_t u = in ? src[j] : -1, v = in ? dst[j] : -1;
    int32_t g = -1;
    if (in && u >= ptr[0] && u < ptr[G]) {
        int64_t lo = 0, hi = G;                               // last g with ptr[g] <= u (ptr is non-decreasing)
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (ptr[mid] <= u) lo = mid; else hi = mid; }
        // empty graphs share their ptr value with the next one: the search ends on the LAST graph starting at or below u,
        // which is the one whose range [ptr[g], ptr[g+1]) can hold u
        if (u >= ptr[lo] && u < ptr[lo + 1] && v >= ptr[lo] && v < ptr[lo + 1]) g = (int32_t)lo;
    }
    if (in) owner[j] = g;
    // per graph: column count, first and last column.  A PyG batch keeps a graph's columns together, so the lanes of a wave hold one
    // or two graphs: one lane per DISTINCT graph of the wave does the three atomics (one per column serialised on the same words:
    // 25 us of this 5-us kernel)
    uint64_t todo = __ballot(g >= 0);
    const int lane = (int)(threadIdx.x & 63);
    while (todo) {
        const int l0 = __ffsll((long long)todo) - 1;
        const int32_t g0 = __shfl(g, l0, 64);
        const uint64_t same = __ballot(g == g0);
        if (lane == l0) {
            const uint32_t jb = (uint32_t)(j - l0);                      // column of lane 0 of this wave
            atomicAdd(&cnt[g0], (uint32_t)__popcll(same));
            atomicMax(&jminc[g0], 0xFFFFFFFFu - (jb + (uint32_t)l0));     // l0 is the lowest lane holding g0
            atomicMax(&jmax[g0], jb + (uint32_t)(63 - __clzll((long long)same)));
        }
        todo &= ~same;
    }
}

struct BpBuild {
    const int64_t *src, *dst, *ptr;
    const int32_t *owner;                 // two-kernel variant: graph of every column (ugs_bp_assign) ...
    const uint32_t *cnt, *jminc, *jmax;   // ... and per graph its column count, 0xFFFFFFFF - first column, last column
    int64_t G, E;
    unsigned long long *bump;    // device counter handing out CSR space: never reset, the host knows its value at launch (bump_base);
    unsigned long long bump_base;//   a graph's entries may lie anywhere in adj / adjf, rowptr is absolute
    uint32_t epoch;              // written to *flag when a graph exceeds the limits (no zeroing between calls)
    // what the host reads (pinned host memory, written by the kernel): keys[G] u64 | cnt[G] | jminc[G] | jmax[G] | flag
    unsigned long long *h_keys;
    uint32_t *h_cnt, *h_jminc, *h_jmax, *h_flag;
    const int64_t *rstart;       // [G] first rowptr entry of graph g (host: prefix of n_g + 1 over the non-degenerate graphs)
    int32_t k;
    int64_t *rowptr;
    int2 *adj, *adjf;
    int32_t *vrank;              // [rows]: rank of vertex x of graph g at rstart[g] + x (read by ugs_bp_roots for graphs the LRU does not know)
    // completion signal for the host: every block counts itself on a device counter (never reset: the host knows its value before the
    // launch), the block that completes the count writes the launch's epoch to a word in pinned host memory -- the host polls that word
    // instead of waiting on the stream (5 us against 11 us for an empty kernel on this stack, tools/sync_probe.hip)
    unsigned long long *done;
    unsigned long long done_target;
    uint32_t *h_done;
};

// every thread of the block calls this at the end of the kernel; the block's writes to pinned host memory are fenced by their writers
__device__ __forceinline__ void bp_signal_done(unsigned long long *done, unsigned long long target, uint32_t *h_done, uint32_t epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        if (atomicAdd(done, 1ull) + 1ull == target) { __threadfence_system(); *(volatile uint32_t *)h_done = epoch; }
    }
}

constexpr int kBpMaxCols = UGS_BATCH_PASS_MAX_COLS, kBpMaxN = UGS_BATCH_PASS_MAX_N;
constexpr int kBpMaskChunks = 512;                         // per wave: batches of up to 131 072 columns keep their ballots

template <bool FUSED>
__device__ __forceinline__ void bp_build_graph(const BpBuild &a) {
    __shared__ uint16_t LU[kBpMaxCols], LV[kBpMaxCols];
    __shared__ int32_t LC[kBpMaxCols];                       // batch column of local column t
    __shared__ uint16_t DEG[kBpMaxN], RNK[kBpMaxN];
    __shared__ uint32_t RP[kBpMaxN + 1], CUR[kBpMaxN];
    __shared__ unsigned long long MSK[kBpMaxN];
    __shared__ uint32_t wsum[kBpBlock / 64];
    __shared__ uint32_t run_sh, cstart_sh;
    __shared__ unsigned long long MKS[FUSED ? kBpBlock / 64 : 1][FUSED ? kBpMaskChunks : 1];   // FUSED: which lanes of which chunk hold a column of this graph
    const int g = (int)blockIdx.x;
    const int64_t lo = a.ptr[g], n64 = a.ptr[g + 1] - lo;
    if (n64 <= 0 || n64 < a.k) {                                  // degenerate: rows of -1, nothing to build, never looked up
        if (threadIdx.x == 0) { a.h_keys[g] = 0ull; a.h_cnt[g] = 0u; a.h_jminc[g] = 0u; a.h_jmax[g] = 0u; }
        return;
    }
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t cn = FUSED ? 0u : a.cnt[g];
    if (n64 > kBpMaxN || cn > (uint32_t)kBpMaxCols) { if (tid == 0) *a.h_flag = a.epoch; return; }
    const int n = (int)n64;
    if (tid == 0) run_sh = 0u;
    __syncthreads();
    // 1. the graph's columns in column order: ordered compaction of the span [first, last] of its columns (FUSED: of all columns --
    //    a column is the graph's iff both endpoints lie in its node range, reference src/ugs_sampler_batch_extension.cpp:52-58)
    if constexpr (FUSED) {
        // every wave takes a contiguous quarter of the columns: count, one block-wide exchange of the four counts, then place -- two
        // passes over the columns (the second from cache) instead of three block barriers per chunk of 256 columns
        const uint32_t E32 = (uint32_t)a.E, per = (((E32 + 3u) / 4u) + 63u) & ~63u;
        const uint32_t w0 = (uint32_t)wv * per, w1 = (w0 + per < E32) ? w0 + per : E32;
        const uint32_t nchunk = w1 > w0 ? (w1 - w0 + 63u) / 64u : 0u;
        const bool keep = nchunk <= (uint32_t)kBpMaskChunks;               // the first pass's ballots are kept: the second touches only chunks with a column
        uint32_t count = 0;
        // four chunks per iteration: eight independent loads in flight (one chunk at a time the loop runs at one memory round trip
        // per 64 columns: 31 us for a 4672-column batch)
        for (uint32_t c0 = 0; c0 < nchunk; c0 += 4) {
            int64_t su[4], sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t j = w0 + (c0 + q) * 64u + (uint32_t)lane;
                su[q] = -1; sv[q] = -1;
                if (c0 + q < nchunk && j < w1) { su[q] = a.src[j] - lo; sv[q] = a.dst[j] - lo; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t mk = __ballot((uint64_t)su[q] < (uint64_t)n64 && (uint64_t)sv[q] < (uint64_t)n64);
                count += (uint32_t)__popcll(mk);
                if (keep && lane == 0 && c0 + q < nchunk) MKS[wv][c0 + q] = mk;
            }
        }
        if (lane == 0) wsum[wv] = count;
        __syncthreads();
        uint32_t off = 0, tot = 0;
        for (int i = 0; i < kBpBlock / 64; ++i) { if (i < wv) off += wsum[i]; tot += wsum[i]; }
        if (tid == 0) run_sh = tot;
        if (tot <= (uint32_t)kBpMaxCols)
            for (uint32_t c = 0; c < nchunk; ++c) {
                uint64_t mk;
                int64_t su = 0, sv = 0;
                const uint32_t j = w0 + c * 64u + (uint32_t)lane;
                if (keep) {
                    mk = MKS[wv][c];
                    if (!mk) continue;                                          // wave-uniform
                    if ((mk >> lane) & 1ull) { su = a.src[j] - lo; sv = a.dst[j] - lo; }
                } else {
                    bool mine = false;
                    if (j < w1) { su = a.src[j] - lo; sv = a.dst[j] - lo; mine = (uint64_t)su < (uint64_t)n64 && (uint64_t)sv < (uint64_t)n64; }
                    mk = __ballot(mine);
                }
                if ((mk >> lane) & 1ull) { const uint32_t t = off + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull)); LU[t] = (uint16_t)su; LV[t] = (uint16_t)sv; LC[t] = (int32_t)j; }
                off += (uint32_t)__popcll(mk);
            }
        __syncthreads();
    } else if (cn) {
        const uint32_t j0 = 0xFFFFFFFFu - a.jminc[g], j1 = a.jmax[g];
        for (uint32_t base = j0; base <= j1; base += kBpBlock) {
            const uint32_t j = base + (uint32_t)tid;
            const bool mine = j <= j1 && a.owner[j] == g;
            int64_t su = 0, sv = 0;
            if (mine) { su = a.src[j] - lo; sv = a.dst[j] - lo; }
            const uint64_t mk = __ballot(mine);
            if (lane == 0) wsum[wv] = (uint32_t)__popcll(mk);
            __syncthreads();
            uint32_t off = run_sh;
            for (int i = 0; i < wv; ++i) off += wsum[i];
            if (mine) {
                const uint32_t t = off + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
                LU[t] = (uint16_t)su; LV[t] = (uint16_t)sv; LC[t] = (int32_t)j;
            }
            __syncthreads();
            if (tid == 0) { uint32_t tot = 0; for (int i = 0; i < kBpBlock / 64; ++i) tot += wsum[i]; run_sh += tot; }
            __syncthreads();
        }
    }
    if constexpr (FUSED) {
        cn = run_sh;
        if (cn > (uint32_t)kBpMaxCols) { if (tid == 0) *a.h_flag = a.epoch; return; }
    }
    // the graph's place in adj / adjf, and what the host needs of it
    if (tid == 0) {
        cstart_sh = (uint32_t)(atomicAdd(a.bump, 2ull * cn) - a.bump_base);
        a.h_cnt[g] = cn;
        a.h_jminc[g] = cn ? 0xFFFFFFFFu - (uint32_t)LC[0] : 0u;
        a.h_jmax[g] = cn ? (uint32_t)LC[cn - 1] : 0u;
    }
    __syncthreads();
    // 2. the LRU key (reference include/cache.hpp:81-109; all columns: cn <= 1000 on this path): one lane, while the others count degrees
    if (wv == kBpBlock / 64 - 1) {   // the whole LAST wave (the vertex loops below start with the first) with wave-uniform values: the chain of
                                     // 64-bit multiplies runs on the scalar unit
        const unsigned long long prime = 1099511628211ull;
        unsigned long long h = 14695981039346656037ull;
        h = (h ^ (unsigned long long)n) * prime;
        h = (h ^ (unsigned long long)cn) * prime;
        for (uint32_t t0 = 0; t0 < cn; t0 += 64) {       // 64 columns per LDS round trip, then lane by lane out of registers (v_readlane)
            const uint32_t tl = t0 + (uint32_t)lane < cn ? t0 + (uint32_t)lane : 0u;
            const int mu = (int)LU[tl], mv = (int)LV[tl];
            const uint32_t cnt64 = cn - t0 < 64u ? cn - t0 : 64u;
            for (uint32_t i = 0; i < cnt64; ++i) {
                const uint32_t uu = (uint32_t)__builtin_amdgcn_readlane(mu, (int)i), vv = (uint32_t)__builtin_amdgcn_readlane(mv, (int)i);
                h = (h ^ (unsigned long long)uu) * prime;
                h = (h ^ (unsigned long long)vv) * prime;
            }
        }
        if (lane == 0) { a.h_keys[g] = h; __threadfence_system(); }
    }
    // 3. degrees of the symmetrised multigraph (a self loop adds two entries to its row: reference src/preproc.cpp:47-60): one LDS
    //    atomic per endpoint
    for (int x = tid; x < n; x += kBpBlock) CUR[x] = 0u;
    __syncthreads();
    for (uint32_t t = tid; t < cn; t += kBpBlock) { atomicAdd(&CUR[LU[t]], 1u); atomicAdd(&CUR[LV[t]], 1u); }
    __syncthreads();
    for (int x = tid; x < n; x += kBpBlock) DEG[x] = (uint16_t)CUR[x];
    __syncthreads();
    // 4. row pointer (exclusive scan of the degrees) and the maximum degree
    {
        uint32_t carry = 0;
        for (int base = 0; base < n; base += kBpBlock) {
            const int x = base + tid;
            const uint32_t d = x < n ? DEG[x] : 0u;
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
    }
    // 5. rank(x) in the reference's vertex order = ascending (CSR degree, vertex id) (src/preproc.cpp:88-140 restated: ugs_host.cpp)
    for (int x = tid; x < n; x += kBpBlock) {
        const uint32_t dx = DEG[x];
        uint32_t r = 0;
        for (int y = 0; y < n; ++y) { const uint32_t dy = DEG[y]; r += (dy < dx || (dy == dx && y < x)) ? 1u : 0u; }
        RNK[x] = (uint16_t)r;
    }
    __syncthreads();
    // 6. the plan's arrays.  Rows hold their entries in COLUMN order -- per column u's row first, then v's, so both entries of a self
    //    loop land in its row in that order (reference src/preproc.cpp:62-86).  The 2*cn endpoints e = 2t + side are placed 64 at a
    //    time by ONE wave, in order: a running cursor per row (CUR), inside a chunk the lanes holding the same row find each other
    //    through a 64-bit member mask per row in LDS (atomic OR, read back) and rank themselves by lane.
    const int64_t abase = (int64_t)cstart_sh, rbase = a.rstart[g];
    for (int x = tid; x <= n; x += kBpBlock) a.rowptr[rbase + x] = abase + (int64_t)RP[x];
    for (int x = tid; x < n; x += kBpBlock) a.vrank[rbase + x] = (int32_t)RNK[x];
    for (int x = tid; x < n; x += kBpBlock) { CUR[x] = RP[x]; MSK[x] = 0ull; }
    __syncthreads();
    if (wv == 0) {
        const uint32_t ne = 2u * cn;
        for (uint32_t e0 = 0; e0 < ne; e0 += 64) {
            const uint32_t e = e0 + (uint32_t)lane;
            const bool on = e < ne;
            const uint32_t t = on ? (e >> 1) : 0u;
            const int u = LU[t], v = LV[t];
            const int row = (e & 1u) ? v : u, other = (e & 1u) ? u : v;       // endpoint 2t: entry v in u's row; endpoint 2t+1: entry u in v's row
            if (on) atomicOr(&MSK[row], 1ull << lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            const unsigned long long mates = on ? MSK[row] : 0ull;
            const uint32_t below = (uint32_t)__popcll(mates & ((1ull << lane) - 1ull));
            const uint32_t cur = on ? CUR[row] : 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (on) {
                const int64_t pp = abase + (int64_t)(cur + below);
                a.adj[pp] = make_int2(other, (int)RNK[other]);
                a.adjf[pp] = make_int2(other, LC[t]);
                if (below == 0u) { CUR[row] = cur + (uint32_t)__popcll(mates); MSK[row] = 0ull; }   // the row's first lane of the chunk moves its cursor on
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
    }
}

template <bool FUSED>
__global__ __launch_bounds__(kBpBlock) void ugs_bp_build(BpBuild a) {
    bp_build_graph<FUSED>(a);
    bp_signal_done(a.done, a.done_target, a.h_done, a.epoch);
}

// ---------------------------------------------------------------------------------------------------------------------------
// ugs_bp_roots: the REST of the reference's preprocessing (src/preproc.cpp:142-256 suffix degrees, k-reachability of every root
// inside its suffix graph, bucket weights max(1, sdeg)^(k-1), Z; include/sampler.hpp:44-69 the Vose alias table) for the graphs
// of a device-built plan that the LRU did not know -- one 256-thread block per such graph, from the CSR and ranks ugs_bp_build has
// just written.  The root records (or the viable list of relaxation levels 1 / 2, src/sampler.cpp:121-150) go straight into the
// device's arena; the host receives 32 bytes per graph (level, list length, Z, degree statistics) and never sees the graph.
// Floating point: every operation whose result the reference's output depends on is done in the reference's order in IEEE fp64
// (this file is built with -ffp-contract=off): b by repeated multiplication, Z summed over the order positions ascending,
// p = b * n / Z, and the alias table by ONE lane -- Vose's loop is sequential by definition (two stacks filled in ascending
// index, p[l] = (p[l] + p[s]) - 1.0); the lane keeps the top of the `large` stack and its p in registers, so an iteration is
// two dependent LDS reads.
// ---------------------------------------------------------------------------------------------------------------------------
struct BpRoots {
    const int64_t *ptr, *rstart, *rowptr;
    const int2 *adj;
    const int32_t *vrank;
    const UgsBpMissIn *in;       // pinned host memory, read by the kernel
    UgsBpMissOut *out;           // pinned host memory, written by the kernel
    UgsRootRec *roots;
    int2 *via;
    int32_t k;
    unsigned long long *done;    // completion signal, as in BpBuild
    unsigned long long done_target;
    uint32_t *h_done;
    uint32_t epoch;
};

constexpr int kBrMaxN = UGS_BATCH_ROOTS_MAX_N;

__global__ __launch_bounds__(kBpBlock) void ugs_bp_roots(BpRoots a) {
    __shared__ double P[kBrMaxN];                                  // bucket weight -> scaled weight -> acceptance probability, in place
    __shared__ uint16_t ORD[kBrMaxN], RNK[kBrMaxN], SDEG[kBrMaxN], ALI[kBrMaxN], LO[kBrMaxN], HI[kBrMaxN];
    __shared__ uint16_t RP[kBrMaxN + 1], NBR[2 * kBpMaxCols];
    __shared__ uint16_t LIST[kBpBlock][UGS_KMAX];                  // per thread: the vertices its root has reached
    __shared__ UgsBpMissIn mi_sh;
    __shared__ unsigned long long s2_sh;
    __shared__ uint32_t maxd_sh, nz_sh;
    __shared__ double z_sh;
    const int tid = (int)threadIdx.x;
    if (tid == 0) { mi_sh = a.in[blockIdx.x]; s2_sh = 0ull; maxd_sh = 0u; }
    __syncthreads();
    const int g = mi_sh.g, k = a.k;
    const int n = (int)(a.ptr[g + 1] - a.ptr[g]);
    const int64_t rb = a.rstart[g], abase = a.rowptr[rb];
    const int nnz = (int)(a.rowptr[rb + n] - abase);
    for (int x = tid; x <= n; x += kBpBlock) RP[x] = (uint16_t)(a.rowptr[rb + x] - abase);
    for (int e = tid; e < nnz; e += kBpBlock) NBR[e] = (uint16_t)a.adj[abase + e].x;
    for (int x = tid; x < n; x += kBpBlock) { const int r = a.vrank[rb + x]; RNK[x] = (uint16_t)r; ORD[r] = (uint16_t)x; ALI[x] = 0; }
    __syncthreads();
    {   // degree statistics for the tier choice (host: order_by_degree): sums of integers, exact in any order
        unsigned long long s2 = 0ull; uint32_t md = 0u;
        for (int x = tid; x < n; x += kBpBlock) { const uint32_t d = (uint32_t)RP[x + 1] - (uint32_t)RP[x]; s2 += (unsigned long long)d * d; md = d > md ? d : md; }
        if (s2) atomicAdd(&s2_sh, s2);
        if (md) atomicMax(&maxd_sh, md);
    }
    // suffix degree, reachability and weight of every order position (host: root_stats_host + weigh_roots)
    uint16_t *L = LIST[tid];
    for (int vi = tid; vi < n; vi += kBpBlock) {
        const int v = ORD[vi];
        int c = 0;
        for (int p = RP[v], e = RP[v + 1]; p < e; ++p) c += (int)RNK[NBR[p]] >= vi ? 1 : 0;
        SDEG[vi] = (uint16_t)c;
        int cnt = 1;
        L[0] = (uint16_t)v;
        if (n <= 64) {                                                   // the reached set as a register bit mask (same visits, same order: the count is all that is used)
            unsigned long long reached = 1ull << v;
            for (int h = 0; h < cnt && cnt < k; ++h) {
                const int u = L[h];
                for (int p = RP[u], e = RP[u + 1]; p < e && cnt < k; ++p) {
                    const int w = NBR[p];
                    if ((int)RNK[w] < vi || ((reached >> w) & 1ull)) continue;
                    reached |= 1ull << w;
                    L[cnt++] = (uint16_t)w;
                }
            }
        } else
        for (int h = 0; h < cnt && cnt < k; ++h) {
            const int u = L[h];
            for (int p = RP[u], e = RP[u + 1]; p < e && cnt < k; ++p) {
                const int w = NBR[p];
                if ((int)RNK[w] < vi) continue;
                bool seen = false;
                for (int i = 0; i < cnt; ++i) seen |= (int)L[i] == w;
                if (!seen) L[cnt++] = (uint16_t)w;
            }
        }
        double b = 0.0;
        if (cnt >= k) {
            const double d = (double)(c > 1 ? c : 1);
            b = 1.0;
            for (int t = 1; t < k; ++t) b *= d;
        }
        P[vi] = b;
    }
    __syncthreads();
    if (tid == 0) {                                                  // Z in order-position order; adding the zeros of unreachable roots changes nothing
        double z = 0.0; uint32_t nz = 0u;
        for (int vi = 0; vi < n; ++vi) { const double b = P[vi]; z += b; nz += b > 0.0 ? 1u : 0u; }
        z_sh = z; nz_sh = nz;
    }
    __syncthreads();
    const double Z = z_sh;
    const uint32_t nz = nz_sh;
    int level = 0, n_via = 0;
    if (Z > 0.0) {
        for (int i = tid; i < n; i += kBpBlock) P[i] = P[i] * n / Z;
        __syncthreads();
        if (tid == 0) {
            int nl = 0, nh = 0;
            for (int i = 0; i < n; ++i) { if (P[i] < 1.0) LO[nl++] = (uint16_t)i; else HI[nh++] = (uint16_t)i; }
            if (nl > 0 && nh > 0) {
                // One lane, a chain of dependent LDS reads per step (stack top, its weight): the entries the NEXT step will need are
                // requested before this step's arithmetic -- the next small one is either the large one just demoted (its weight is in a
                // register) or the entry below the top, the next large one the entry below the current.  Same reads, same arithmetic,
                // same order of operations as the host's build_alias (reference include/sampler.hpp:44-69).
                int l = HI[nh - 1];
                double pl = P[l];
                int s = LO[nl - 1];
                double ps = P[s];
                while (nl > 0 && nh > 0) {
                    --nl;
                    const int s_below = nl > 0 ? LO[nl - 1] : 0;
                    const int l_below = nh > 1 ? HI[nh - 2] : 0;
                    const double ps_below = P[s_below], pl_below = P[l_below];
                    ALI[s] = (uint16_t)l;
                    pl = (pl + ps) - 1.0;
                    if (pl < 1.0) {
                        P[l] = pl;
                        --nh;
                        LO[nl++] = (uint16_t)l;
                        s = l; ps = pl;                                  // popped next: the demoted one
                        l = l_below; pl = pl_below;                      // (unused when nh == 0)
                    } else { s = s_below; ps = ps_below; }
                }
            }
            for (int i = 0; i < nh; ++i) P[HI[i]] = 1.0;              // leftovers of either stack accept with probability 1
            for (int i = 0; i < nl; ++i) P[LO[i]] = 1.0;
        }
        __syncthreads();
    }
    if (nz > 0u) {
        UgsRootRec *rec = a.roots + mi_sh.roots_off;
        for (int vi = tid; vi < n; vi += kBpBlock) {
            UgsRootRec r;
            r.prob = P[vi]; r.alias = (int32_t)ALI[vi]; r.v_self = (int32_t)ORD[vi]; r.v_alias = (int32_t)ORD[ALI[vi]]; r.pad = 0;
            rec[vi] = r;
        }
    } else if (tid == 0) {                                           // relaxed root draw: the roots with a suffix neighbour, else all of them
        int2 *via = a.via + mi_sh.via_off;
        level = 1;
        for (int vi = 0; vi < n; ++vi) if (SDEG[vi] > 0) via[n_via++] = make_int2(vi, (int)ORD[vi]);
        if (n_via == 0) { level = 2; for (int vi = 0; vi < n; ++vi) via[n_via++] = make_int2(vi, (int)ORD[vi]); }
    }
    if (tid == 0) {
        UgsBpMissOut o;
        o.level = level; o.n_viable = n_via; o.nonzero = (int32_t)nz; o.max_deg = (int32_t)maxd_sh;
        o.Z = Z;
        o.sb_deg = nnz > 0 ? (double)s2_sh / (double)nnz : 0.0;
        a.out[blockIdx.x] = o;
    }
    bp_signal_done(a.done, a.done_target, a.h_done, a.epoch);
}

}  // namespace

hipError_t ugs_launch_batch_roots(const int64_t *d_ptr, const int64_t *d_rstart, const int64_t *d_rowptr, const int2 *d_adj, const int32_t *d_vrank,
                                  const UgsBpMissIn *h_in, UgsBpMissOut *h_out, int64_t misses, int k, UgsRootRec *d_roots, int2 *d_via,
                                  unsigned long long *d_done, unsigned long long done_base, uint32_t *h_done, uint32_t epoch, hipStream_t s) {
    if (misses <= 0) return hipSuccess;
    BpRoots a{};
    a.ptr = d_ptr; a.rstart = d_rstart; a.rowptr = d_rowptr; a.adj = d_adj; a.vrank = d_vrank; a.in = h_in; a.out = h_out;
    a.roots = d_roots; a.via = d_via; a.k = k;
    a.done = d_done; a.done_target = done_base + (unsigned long long)misses; a.h_done = h_done; a.epoch = epoch;
    hipLaunchKernelGGL(ugs_bp_roots, dim3((unsigned)misses), dim3(kBpBlock), 0, s, a);
    return hipGetLastError();
}

int64_t ugs_batch_pass_fused_work() {
    if (const char *e = std::getenv("UGS_BP_FUSED_WORK")) return std::atoll(e);
    return UGS_BATCH_PASS_FUSED_WORK;
}

hipError_t ugs_launch_batch_pass(const int64_t *d_src, const int64_t *d_dst, int64_t E, const int64_t *d_ptr, int64_t G, int k,
                                 int32_t *d_owner, uint32_t *d_cnt_jminc_jmax /* [3G], two-kernel variant only */, const int64_t *d_rstart,
                                 int64_t *d_rowptr, int2 *d_adj, int2 *d_adjf, int32_t *d_vrank, unsigned long long *d_bump, unsigned long long bump_base,
                                 uint32_t epoch, void *h_back /* pinned: keys[G] u64 | cnt[G] | jminc[G] | jmax[G] | flag | done */,
                                 unsigned long long *d_done, unsigned long long done_base, hipStream_t s) {
    if (G <= 0) return hipSuccess;
    BpBuild a{};
    a.src = d_src; a.dst = d_dst; a.ptr = d_ptr; a.G = G; a.E = E; a.rstart = d_rstart; a.k = k; a.rowptr = d_rowptr; a.adj = d_adj; a.adjf = d_adjf;
    a.vrank = d_vrank;
    a.bump = d_bump; a.bump_base = bump_base; a.epoch = epoch;
    a.h_keys = static_cast<unsigned long long *>(h_back);
    a.h_cnt = reinterpret_cast<uint32_t *>(a.h_keys + G); a.h_jminc = a.h_cnt + G; a.h_jmax = a.h_jminc + G; a.h_flag = a.h_jmax + G;
    a.h_done = a.h_flag + 1;
    a.done = d_done; a.done_target = done_base + (unsigned long long)G;
    if (G * E <= ugs_batch_pass_fused_work()) {            // every block scans every column: no assign launch, no memset, no atomics
        hipLaunchKernelGGL(ugs_bp_build<true>, dim3((unsigned)G), dim3(kBpBlock), 0, s, a);
        return hipGetLastError();
    }
    uint32_t *d_cnt = d_cnt_jminc_jmax, *d_jminc = d_cnt + G, *d_jmax = d_jminc + G;
    hipError_t e = hipMemsetAsync(d_cnt, 0, (size_t)(3 * G) * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ugs_bp_assign, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, d_src, d_dst, E, d_ptr, G, d_owner, d_cnt, d_jminc, d_jmax);
    a.owner = d_owner; a.cnt = d_cnt; a.jminc = d_jminc; a.jmax = d_jmax;
    hipLaunchKernelGGL(ugs_bp_build<false>, dim3((unsigned)G), dim3(kBpBlock), 0, s, a);
    return hipGetLastError();
}
