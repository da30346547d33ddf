// ugs_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the uniform k-subgraph sampler.
//
// What is computed (behavioural contract = the reference `ugs_sampler`, AniruddhaMandal/SS-GNN):
//   walk kernel : per result row b = g*m + i -- xorshift64* stream seeded seed + i*0x9e3779b97f4a7c15
//                 (reference include/sampler.hpp:26-36, src/sampler.cpp:160), root draw through the alias table or
//                 the relaxed viable list (src/sampler.cpp:163-173), k-1 growth steps (rand_grow, src/sampler.cpp:36-85),
//                 per-row induced-edge COUNT; rows come out as nodes[row,k] (-1 padded)
//   scan kernels: exclusive prefix sum of the counts -> edge_ptr
//   fill kernels: induced edges of every complete row in the reference's order (vertex j, then CSR position p;
//                 src/sampler.cpp:232-243) with the endpoint numbering of the requested mode (src/sampler.cpp:258-281)
//
// How (MI355X-first, nothing translated from the reference's std::unordered_set code):
//   * a walk is owned by a GROUP of GS lanes of one 64-lane wavefront (GS = 8 for TU-sized graphs: 8 walks per
//     wave; GS = 64 for large graphs: the wave reads a whole adjacency row with one coalesced 8-byte-per-lane load).
//   * all per-walk state (candidate list D, membership hash, ordered prefixes, bucket table) lives in LDS; groups are
//     independent, so there is no __syncthreads anywhere -- lanes of a group run in lock-step inside their wave.
//   * the reference picks `cut[rng % |cut|]` where `cut` is the ITERATION ORDER of a libstdc++ unordered_set<int>
//     rebuilt at every step.  That order is reproduced without any linked list: D keeps the distinct candidates in
//     first-insertion order (incrementally: drop the chosen vertex, append the new vertex's unseen neighbours), and
//     the container's order is a staged stable grouping computed data-parallel per stage of the bucket chain
//     13 -> 29 -> 59 -> ... with every element in registers: atomicMin + atomicAdd on one LDS word per bucket give its
//     first arrival and size, one DPP wave scan of the sizes over the bucket leaders gives every bucket's start, an
//     atomicAdd-with-return hands out arrival slots into tiny per-bucket position lists from which every element counts
//     the members above it; rank = start + count (stage_mat).  Every stage keeps its own order array -- 16-bit POSITIONS in
//     D, so the final stage names the chosen candidate's position and D is never searched -- valid across growth steps while
//     the removed candidate lies behind the prefix it covers.  The last stage materialises nothing:
//     the bucket holding position rng % |cut| is found by the scan and the element inside it by ballots (stage_final).
//     Materialising stages of up to 128 elements and final stages of up to 64 (one walk per wave) rank every element in
//     registers: a member mask per bucket in LDS (one atomic OR, one read) gives every lane the mask of its bucket-mates and
//     popcounts do the rest (rank_in_registers, mates2_by_table).  In the LDS tiers the stage index is a template parameter,
//     so the chain constants are immediates (mat_at / final_at).
//     (The global-memory fallback tier keeps the simpler "peel round" formulation, select_in_order.)
//   * the neighbour's order rank is stored next to the neighbour id in HBM (int2 adjacency), so the suffix filter is
//     free; root records pack the alias row and both candidate root vertices in 24 bytes; the one-walk-per-wave tiers read
//     a vertex's row from a 128-byte-aligned padded block at an address computed from the vertex (header + first entries:
//     one dependent memory round trip per growth step, scan_prow) -- ugs_device.h.
//   * membership (seen / in sample) is one open-addressing table per walk in LDS, double hashing, inserted with atomicCAS by
//     all lanes of a chunk at once (scan_chunk).
//   * ballot + popcount prefix sums compact new candidates into D and (row-reading fill kernel) edges into the output.
//   * the 64-lane tiers stage the induced edges they meet while scanning rows (stage_hits / stage_flush): the fill kernel
//     of those rows is a plain expand (ugs_fill_staged); rows that do not fit are listed for the row-reading ugs_fill.
#include "ugs_device.h"
#include <cstdlib>

#define UGS_ALIGNED16 __attribute__((aligned(16)))

// Diagnostic build only (-DUGS_STAMPS, never shipped): per-phase shader-cycle totals of the walk kernel.  Lane 0 of every group
// adds the s_memtime difference of each phase to a per-block LDS array (no waits are inserted: a phase is charged the issue
// time and the stalls that really happen inside it), flushed to a buffer of its own (ugs_stamp_buffer) at the end of the block;
// no output value depends on them.  Slots: 0 root + hash reset, 1 scan_row, 2 select (rest), 3 shift + bookkeeping, 4 draw,
// 5 row output + edge staging, 10/11/12 materialised stage (1 / 2 / more elements per lane), 13/14/15 final stage (same split);
// slot 16 + i = executions of slot i.
#ifdef UGS_STAMPS
__device__ unsigned long long ugs_stamp_buffer[32];
#define ST_ ws.ST
#define STAMP_DECL unsigned long long st_t0 = 0
#define STAMP_BEGIN() do { __builtin_amdgcn_sched_barrier(0); st_t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_END(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) { atomicAdd(&ST_[i], t1_ - st_t0); atomicAdd(&ST_[16 + (i)], 1ull); } st_t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
// a routine inside the select phase: its time moves from slot 2 to slot i
#define STAMP_SUB_BEGIN() __builtin_amdgcn_sched_barrier(0); const unsigned long long sti_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define STAMP_SUB_END_OF(parent, i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) { atomicAdd(&ST_[i], t1_ - sti_); atomicAdd(&ST_[parent], sti_ - t1_); atomicAdd(&ST_[16 + (i)], 1ull); } __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_SUB_END(i) STAMP_SUB_END_OF(2, i)
#else
#define STAMP_DECL
#define STAMP_SUB_BEGIN() do {} while (0)
#define STAMP_SUB_END(i) do {} while (0)
#define STAMP_SUB_END_OF(parent, i) do {} while (0)
#define STAMP_BEGIN() do {} while (0)
#define STAMP_END(i) do {} while (0)
#endif

namespace {

// ------------------------------------------------------------------------------------------------------------------
// bucket chain of a libstdc++ unordered_set grown from empty by single inserts (max load factor 1), with the
// multiply-shift constants of an exact x / B for 0 <= x < 2^31:  q = umulhi(x, M) >> (S - 1),
// M = floor(2^(31+S) / B) + 1, S = ceil(log2 B).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kChainLen = 27;
constexpr uint32_t kChainHost[kChainLen] = {13u, 29u, 59u, 127u, 257u, 541u, 1109u, 2357u, 5087u, 10273u, 20753u, 42043u,
    85229u, 172933u, 351061u, 712697u, 1447153u, 2938679u, 5967347u, 12117689u, 24607243u, 49969847u, 101473717u,
    206062531u, 418451333u, 849749479u, 1725587117u};

constexpr int clog2(uint32_t b) { int s = 0; while ((1ull << s) < b) ++s; return s; }
constexpr uint32_t cmagic(uint32_t b) { return (uint32_t)(((1ull << (31 + clog2(b))) / b) + 1ull); }

// O[i] = word offset of stage i's materialised order (every stage keeps its own array so that the orders of the leading
// stages survive from one growth step to the next; sizes padded to even)
struct ChainTab { uint32_t B[kChainLen]; uint32_t M[kChainLen]; uint32_t S[kChainLen]; uint32_t O[kChainLen]; };
constexpr uint32_t ord_words_before(int stage) {
    unsigned long long o = 0;
    for (int i = 0; i < stage && i < kChainLen; ++i) o += (kChainHost[i] + 1u) & ~1u;
    return o > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)o;
}
constexpr ChainTab make_chain() {
    ChainTab t{};
    for (int i = 0; i < kChainLen; ++i) {
        t.B[i] = kChainHost[i]; t.M[i] = cmagic(kChainHost[i]); t.S[i] = (uint32_t)clog2(kChainHost[i]) - 1u;
        t.O[i] = ord_words_before(i);
    }
    return t;
}
__constant__ ChainTab d_chain = make_chain();

__device__ __forceinline__ uint32_t mod_magic(uint32_t x, uint32_t B, uint32_t M, uint32_t S) {
    uint32_t q = __umulhi(x, M) >> S;
    return x - q * B;
}
__device__ __forceinline__ uint32_t mod_stage(uint32_t x, uint32_t B, uint32_t M, uint32_t S) { return mod_magic(x, B, M, S); }

// x mod c for a 64-bit x and a small divisor (c < 2^16), exact, in 32-bit operations: with hi/lo the halves of x,
// x mod c = ((hi mod c) * (2^32 mod c) + lo mod c) mod c, every product below 2^32.  With M = floor((2^32 - 1) / c),
// floor(y * M / 2^32) is floor(y / c) or one less (2^32 - c M <= c, so the estimate falls short by y c / (c 2^32) < 1): one
// correction.  The generic 64-bit remainder costs ~150 instructions per draw.
__device__ __forceinline__ uint32_t mod_small(uint32_t y, uint32_t c, uint32_t M) {
    const uint32_t r = y - __umulhi(y, M) * c;
    const uint32_t r2 = r - c;                          // wraps when r < c: the minimum is the corrected remainder
    return r < r2 ? r : r2;
}
__device__ __forceinline__ uint32_t mod64_with(uint64_t x, uint32_t c, uint32_t M, uint32_t r32 /* 2^32 mod c */) {
    const uint32_t hm = mod_small((uint32_t)(x >> 32), c, M), lm = mod_small((uint32_t)x, c, M);
    return mod_small(hm * r32 + lm, c, M);
}
// (M, 2^32 mod c) for c <= 2048: with one walk per wave the candidate count is wave-uniform, so the per-step draw fetches its
// constants with one scalar load instead of a 32-bit division (entry 0 is never used)
constexpr int kRecipMax = 2048;
struct RecipTab { uint2 e[kRecipMax + 1]; };
constexpr RecipTab make_recip() {
    RecipTab t{};
    t.e[0] = uint2{0u, 0u};
    for (uint32_t c = 1; c <= (uint32_t)kRecipMax; ++c) {
        const uint32_t M = 0xFFFFFFFFu / c;
        uint32_t r32 = 0xFFFFFFFFu - M * c + 1u;
        if (r32 == c) r32 = 0u;
        t.e[c] = uint2{M, r32};
    }
    return t;
}
__constant__ RecipTab d_recip = make_recip();

// SMALL: the caller guarantees c < 2^16 (every LDS tier: c <= its candidate capacity); TABLE: c <= kRecipMax and wave-uniform
template <bool SMALL, bool TABLE = false>
__device__ __forceinline__ uint32_t mod64_by(uint64_t x, uint32_t c) {
    if constexpr (TABLE) {
        const uint2 e = d_recip.e[c];
        return mod64_with(x, c, e.x, e.y);
    }
    if constexpr (!SMALL) { if (c >= 65536u) return (uint32_t)(x % (uint64_t)c); }
    const uint32_t M = 0xFFFFFFFFu / c;
    uint32_t r32 = 0xFFFFFFFFu - M * c + 1u;            // (2^32 - 1) mod c + 1, in [1, c]
    r32 = r32 == c ? 0u : r32;                          // 2^32 mod c
    return mod64_with(x, c, M, r32);
}

// x mod n for the root draw (n = vertices of the graph, < 2^31).  Small graphs take the exact 32-bit route above; for n >= 2^16
// the quotient is estimated in double precision -- x and x / n are each rounded once (relative error 2^-53 each, x / n < 2^48),
// so the estimate is off by at most one -- and the remainder is corrected in integers (two corrections each way).  A third of
// the generic 64-bit remainder's instructions.
__device__ __forceinline__ uint32_t mod64_root(uint64_t x, uint32_t n) {
    if (n < 65536u) return mod64_by<true>(x, n);
    const double q_est = floor((double)x / (double)n);
    const uint64_t q = (uint64_t)q_est;
    int64_t r = (int64_t)(x - q * (uint64_t)n);        // exact modulo 2^64; the true value lies in (-2n, 3n)
    r = r < 0 ? r + (int64_t)n : r;
    r = r < 0 ? r + (int64_t)n : r;
    r = r >= (int64_t)n ? r - (int64_t)n : r;
    r = r >= (int64_t)n ? r - (int64_t)n : r;
    return (uint32_t)r;
}

// xorshift64* (reference include/sampler.hpp:26-36)
struct Rng {
    uint64_t s;
    __device__ __forceinline__ void init(uint64_t seed) {
        s = seed ? seed : 1ull;
    }
    __device__ __forceinline__ uint64_t next() {
        uint64_t x = s;
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        s = x;
        return x * 2685821657736338717ull;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// sub-wave group of GS lanes
// ------------------------------------------------------------------------------------------------------------------
template <int GS> struct Grp {
    int lane;    // 0..GS-1
    int gbase;   // first wave lane of this group
    uint32_t chain = 0u;   // one walk per wave: lane i holds chain value i (chain_lane_const), set by the walk kernel
    __device__ __forceinline__ void init() {
        int wl = (int)(threadIdx.x & 63);
        lane = wl & (GS - 1);
        gbase = wl & ~(GS - 1);
    }
    __device__ __forceinline__ uint64_t ballot(bool p) const {
        uint64_t m = __ballot(p);
        if (GS == 64) return m;
        return (m >> gbase) & ((1ull << (GS & 63)) - 1ull);
    }
    __device__ __forceinline__ bool any(bool p) const { return ballot(p) != 0ull; }
    __device__ __forceinline__ uint64_t lt_mask() const { return (1ull << lane) - 1ull; }
    // number of set bits of a group ballot below this lane.  One walk per wave: v_mbcnt_lo/hi count the bits below the lane in
    // two instructions (the shift-mask-popcount form takes five)
    __device__ __forceinline__ uint32_t below(uint64_t m) const {
        if (GS == 64) return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        return (uint32_t)__popcll(m & lt_mask());
    }
    // value of lane `src` of the group.  One walk per wave: `src` is wave-uniform, so v_readlane puts the result in a scalar
    // register and everything derived from it (row bounds, loop limits, hash slot of the chosen vertex) stays scalar.
    __device__ __forceinline__ uint32_t bcast(uint32_t v, int src) const {
        if (GS == 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, src);
        return __shfl(v, gbase + src, 64);
    }
    // inclusive prefix sum over the group's lanes (sum of x over lanes <= lane).  GS == 64: the whole wave is active
    // here, so the DPP row-shift / row-broadcast scan is used (6 VALU ops, no LDS crossbar traffic).
    __device__ __forceinline__ uint32_t prefix_incl(uint32_t x) const {
        if (GS == 64) {
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
            return x;
        }
        if (GS == 32) {   // both 32-lane halves of the wave are active together (two walks per wave run in lock-step)
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
            return x;
        }
#pragma unroll
        for (int d = 1; d < GS; d <<= 1) {
            uint32_t y = __shfl_up(x, d, GS);
            if (lane >= d) x += y;
        }
        return x;
    }
    // the same when only the first N lanes' sums are needed (GS == 64): 16 lanes are one DPP row -- no row broadcasts --, 32 need one
    template <int N> __device__ __forceinline__ uint32_t prefix_incl_first(uint32_t x) const {
        if (GS != 64 || N > 32) return prefix_incl(x);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
        if (N > 16) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
        return x;
    }
    // group-uniform values: with one walk per wave they are wave-uniform and belong in scalar registers
    __device__ __forceinline__ uint32_t uni(uint32_t x) const { return GS == 64 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)x) : x; }
    __device__ __forceinline__ int64_t uni(int64_t x) const {
        if (GS != 64) return x;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)x >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    }
    __device__ __forceinline__ uint32_t last(uint32_t incl) const {      // value held by the group's last lane
        if (GS == 64) return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        return bcast(incl, GS - 1);
    }
    // inclusive suffix sum over the group's lanes: sum of x over lanes >= lane
    __device__ __forceinline__ uint32_t suffix_incl(uint32_t x) const {
#pragma unroll
        for (int d = 1; d < GS; d <<= 1) {
            uint32_t y = __shfl_down(x, d, GS);
            if (lane + d < GS) x += y;
        }
        return x;
    }
};

// memory-space policies of the per-walk workspace --------------------------------------------------------------------
struct LdsSpace {       // LDS: one wave's LDS operations are serviced in issue order; only the compiler must be fenced
    using TW = uint32_t; using TA = uint16_t;
    using TO = uint16_t;                       // stage orders hold POSITIONS in D (the key is one more LDS read away)
    static __device__ __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
};
struct GlbSpace {       // global-memory fallback: agent-scope fence between phases (rare, correctness-first tier)
    using TW = unsigned long long; using TA = uint32_t;
    using TO = uint32_t;                       // stage orders hold the vertices themselves
    static constexpr int SH = 32;
    static constexpr TW PMASK = 0xFFFFFFFFull, FLAG = 1ull << 63;
    static constexpr TA UNASSIGNED = 0xFFFFFFFFu;
    static __device__ __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); __builtin_amdgcn_wave_barrier(); }
};

template <class SP> struct Work {
    uint32_t *D;             // distinct candidates, first-insertion order           [cap]
    typename SP::TO *ORD;    // materialised order of every non-final stage, stage i at ORD + d_chain.O[i] (LDS tiers: positions in
                             // D, 16 bits each -- a stage's order stays valid while the removed candidate lies behind its prefix, and
                             // so do the positions; global-memory tier: the vertices)
    typename SP::TA *AUX;    // per element: position inside its bucket (generic path only)   [cap]
    typename SP::TW *TBL;    // per bucket: (round|pos) -> (size|first) -> start     [bcap]
    uint32_t *HK;            // membership hash (open addressing): vertex id | kFresh | kInS               [hs]
    uint32_t cap, hmask, hlimit;
#ifdef UGS_STAMPS
    unsigned long long *ST;  // per-block phase totals (diagnostic build)
#endif
};

constexpr uint32_t kEmpty = 0xFFFFFFFFu;

struct Pick { uint32_t w, q; };   // chosen vertex and its position in D (q = c: not known, the caller searches D)

__device__ __forceinline__ uint32_t hash_slot(uint32_t w, uint32_t mask) { return (w * 2654435761u >> 7) & mask; }
// probe sequence slot, slot + step, slot + 2 step, ... (mod table size, a power of two): the step is odd, so the sequence visits
// every slot, and it depends on the key (double hashing), so keys that collide do not queue up behind one another the way they
// do with step 1 -- with 40-odd lanes inserting at once it is the LONGEST probe sequence of a chunk that costs
// (measured on C5 against step 1: 6.88 -> 6.81 ms per 1M walks)
__device__ __forceinline__ uint32_t hash_step(uint32_t w) { return ((w * 2246822519u) >> 20) | 1u; }

// iteration-order selection: vertex at position `rsel` of the libstdc++ unordered_set<int> built by inserting
// D[0..c) one by one (see file header).  c >= 1, rsel < c.  Group-uniform result.
template <int GS, class SP>
__device__ __forceinline__ uint32_t select_in_order(const Work<SP> &ws, const Grp<GS> &g, uint32_t c, uint32_t rsel, int &nvalid) {
    using TW = typename SP::TW;
    using TA = typename SP::TA;
    int fs = 0;
    while (fs < kChainLen - 1 && d_chain.B[fs] < c) ++fs;                // the final stage: first chain value >= c
    for (int stage = nvalid < fs ? nvalid : fs; stage < kChainLen; ++stage) {
        const uint32_t B = d_chain.B[stage], M = d_chain.M[stage], S = d_chain.S[stage];
        const uint32_t L = c < B ? c : B;
        const bool final = stage == fs;
        const uint32_t *OLD = stage ? ws.ORD + d_chain.O[stage - 1] : ws.D;
        uint32_t *NEW = ws.ORD + d_chain.O[stage];
        const uint32_t n_old = stage ? d_chain.B[stage - 1] : 0u;
        for (uint32_t b = g.lane; b < B; b += GS) ws.TBL[b] = 0;
        for (uint32_t t = g.lane; t < L; t += GS) ws.AUX[t] = SP::UNASSIGNED;
        SP::sync();
        // peel rounds: in round r the still-unplaced element with the largest position wins its bucket
        for (uint32_t round = 1; round <= L; ++round) {        // at most (largest bucket size) <= L rounds
            for (uint32_t t = g.lane; t < L; t += GS) {
                if (ws.AUX[t] == SP::UNASSIGNED) {
                    uint32_t key = (t < n_old) ? OLD[t] : ws.D[t];
                    uint32_t b = mod_magic(key, B, M, S);
                    atomicMax(&ws.TBL[b], ((TW)round << SP::SH) | (TW)t);
                }
            }
            SP::sync();
            bool left = false;
            for (uint32_t t = g.lane; t < L; t += GS) {
                if (ws.AUX[t] == SP::UNASSIGNED) {
                    uint32_t key = (t < n_old) ? OLD[t] : ws.D[t];
                    uint32_t b = mod_magic(key, B, M, S);
                    if ((uint32_t)(ws.TBL[b] & SP::PMASK) == t) ws.AUX[t] = (TA)(round - 1);
                    else left = true;
                }
            }
            SP::sync();
            if (!g.any(left)) break;
        }
        // now TBL[b] = (bucket size << SH) | first position.  Buckets are laid out by DESCENDING first position:
        // suffix scan of the sizes over the positions, highest chunk first; the leader converts its bucket entry to the
        // bucket's start rank (it is the last element of its bucket to be visited).
        uint32_t carry = 0;
        const int nchunks = (int)((L + GS - 1) / GS);
        for (int ch = nchunks - 1; ch >= 0; --ch) {
            const uint32_t t = (uint32_t)ch * GS + g.lane;
            const bool in = t < L;
            uint32_t b = 0, gsz = 0;
            bool leader = false;
            if (in) {
                uint32_t key = (t < n_old) ? OLD[t] : ws.D[t];
                b = mod_magic(key, B, M, S);
                TW v = ws.TBL[b];
                leader = !(v & SP::FLAG) && (uint32_t)(v & SP::PMASK) == t;
                gsz = leader ? (uint32_t)(v >> SP::SH) : 0u;
            }
            uint32_t incl = g.suffix_incl(gsz);
            uint32_t excl = incl - gsz + carry;
            carry += g.bcast(incl, 0);
            if (leader) ws.TBL[b] = SP::FLAG | (TW)excl;
            SP::sync();
        }
        // rank = bucket start + position inside the bucket
        bool have = false;
        uint32_t mine = 0;
        for (uint32_t t = g.lane; t < L; t += GS) {
            uint32_t key = (t < n_old) ? OLD[t] : ws.D[t];
            uint32_t b = mod_magic(key, B, M, S);
            uint32_t rank = (uint32_t)(ws.TBL[b] & ~SP::FLAG) + (uint32_t)ws.AUX[t];
            if (final) { if (rank == rsel) { have = true; mine = key; } }
            else NEW[rank] = key;
        }
        SP::sync();
        if (final) {
            uint64_t mk = g.ballot(have);
            int src = mk ? (__ffsll((long long)mk) - 1) : 0;
            return g.bcast(mine, src);
        }
        nvalid = stage + 1;
    }
    return ws.D[0];   // unreachable for c within the chain
}

// One NON-FINAL stage of the order computation (the table is full: B elements into B buckets) with every element held in
// REGISTERS: lane l owns the NJ consecutive positions [l*NJ, l*NJ + NJ) (NJ odd -> the lane stride is conflict-free on
// the 32 LDS banks); the LDS operations of a phase are issued back to back and waited for once.
//   1. atomicMin  -> first position of every bucket           2. atomicAdd -> its size (same word: size<<16 | first)
//   3. ONE group-wide DPP scan of the sizes over the bucket leaders (buckets are laid out by DESCENDING first position)
//      -> leaders store their bucket's start rank
//   4. atomicAdd-with-return hands every element the bucket's start and an arrival slot; the element's position goes to
//      a uint16 scratch list at start + slot (behind the bucket table)
//   5. every element reads its bucket's (tiny) list and counts the members above it -> rank = start + count; scatter.
// No loop over rounds: nine LDS round trips whatever the bucket sizes (a bucket of more than UNR members takes a short
// extra loop).
template <int GS, int NJ, bool POS_IN_NEW = false>
__device__ __forceinline__ void stage_mat(const Work<LdsSpace> &ws, const Grp<GS> &g, const uint16_t *OLD, uint16_t *NEW,
                                          uint32_t n_old, uint32_t B, uint32_t M, uint32_t S) {
    constexpr int UNR = NJ <= 5 ? 6 : 4;
    const uint32_t t0 = (uint32_t)g.lane * NJ;
    uint32_t pos[NJ], bk[NJ];                                                   // position in D; the key is only needed for the bucket
    bool valid[NJ];
    {
        uint4 *T4 = reinterpret_cast<uint4 *>(ws.TBL);
        const uint32_t n4 = (B + 3u) >> 2;
        for (uint32_t i = g.lane; i < n4; i += GS) T4[i] = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const uint32_t t = t0 + j;
        valid[j] = t < B;
        pos[j] = t;
        if (valid[j] && t < n_old) pos[j] = OLD[t];
    }
    uint32_t key[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { key[j] = 0u; if (valid[j]) key[j] = ws.D[pos[j]]; }
    LdsSpace::sync();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        bk[j] = mod_stage(key[j], B, M, S);
        atomicMin(&ws.TBL[bk[j]], valid[j] ? t0 + j : 0xFFFFu);
    }
    LdsSpace::sync();
#pragma unroll
    for (int j = 0; j < NJ; ++j) atomicAdd(&ws.TBL[bk[j]], valid[j] ? 0x10000u : 0u);
    LdsSpace::sync();
    uint32_t cnt[NJ], gs[NJ], mine_total = 0u;
#pragma unroll
    for (int j = 0; j < NJ; ++j) gs[j] = ws.TBL[bk[j]];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        cnt[j] = valid[j] ? (gs[j] >> 16) : 0u;
        gs[j] = (valid[j] && (gs[j] & 0xFFFFu) == t0 + j) ? cnt[j] : 0u;      // size, if this element leads its bucket
        mine_total += gs[j];
    }
    const uint32_t incl = g.prefix_incl(mine_total);
    // ranks taken by buckets led from higher lanes.  The sizes of all buckets add up to the stage's B elements: no read of the last
    // lane's sum (a v_readlane behind a DPP step and in front of its first use costs two wait-state fillers; an s_nop is 5-8 wave-cycles)
    uint32_t run = B - incl;
    LdsSpace::sync();                                                          // every lane has read the sizes
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
        if (gs[j]) ws.TBL[bk[j]] = run;                                        // arrivals 0 | bucket start
        run += gs[j];
    }
    LdsSpace::sync();
    // the per-bucket position lists: behind the table, or -- in a tier whose table is only as large as its largest stage -- in the
    // stage's own order array, which is written only after the last read of the lists (the barrier in front of the scatter)
    uint16_t *POS = POS_IN_NEW ? NEW : reinterpret_cast<uint16_t *>(ws.TBL + ((B + 3u) & ~3u));
    uint32_t st[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) st[j] = atomicAdd(&ws.TBL[bk[j]], valid[j] ? 0x10000u : 0u);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const uint32_t slot = st[j] >> 16;
        st[j] &= 0xFFFFu;
        if (valid[j]) POS[st[j] + slot] = (uint16_t)(t0 + j);
    }
    LdsSpace::sync();
    uint32_t rho[NJ];
    bool more = false;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        uint32_t pv[UNR];
#pragma unroll
        // reads past the list stay inside the table.  (Reading only the entries that exist -- fewer active lanes per LDS read, more
        // instructions -- measured +4 % slower on the degree-160 job, neutral on degree 80: profiles/r04_ab_small_tier_probes.txt.)
        for (int i = 0; i < UNR; ++i) pv[i] = POS[st[j] + i];
        rho[j] = 0u;
#pragma unroll
        for (int i = 0; i < UNR; ++i) rho[j] += ((uint32_t)i < cnt[j] && pv[i] > t0 + j) ? 1u : 0u;
        more = more || cnt[j] > (uint32_t)UNR;
    }
    if (g.any(more)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            for (uint32_t i = UNR; i < cnt[j]; ++i) rho[j] += (POS[st[j] + i] > t0 + j) ? 1u : 0u;
    }
    LdsSpace::sync();
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (valid[j]) NEW[st[j] + rho[j]] = (uint16_t)pos[j];
    LdsSpace::sync();
}

// Stages of at most 64 elements with one walk per wave: ONE element per lane.  Every lane gets the 64-bit mask of the lanes
// that share its bucket (through a member mask per bucket in LDS, see rank_in_registers);
// popcounts of that mask give the bucket's size, its first position (the leader) and the number of members above the lane,
// one DPP scan of the sizes over the leaders gives every bucket's start and one ds_bpermute fetches the leader's start.
// rank = start + members above -- the same definition as stage_mat.

template <int MAXN = 64>      // MAXN: compile-time bound on the valid lanes (the 13- and 29-element stages scan one or two DPP rows only)
__device__ __forceinline__ uint32_t rank_in_registers(const Grp<64> &g, bool valid, uint32_t bk, uint32_t *TBL, uint32_t B, uint32_t n_valid) {
    // mask of the lane's bucket-mates through the (otherwise idle) bucket table: one 64-bit member mask per bucket, set by an
    // LDS atomic OR and read back -- one LDS round trip in place of a radix pass of ballots over the bits of the bucket number
    // (4 vector instructions per bit: 62-78 -> 37-41 per stage).  B <= 127: a final of <= 64 candidates may have 127 buckets.
    // (Neutral while the kernel still spent its time elsewhere, -3.3 % once the order stages dominated: 5.45 -> 5.27 ms.)
    unsigned long long *T8 = reinterpret_cast<unsigned long long *>(TBL);
    T8[g.lane] = 0ull;                                                       // all 64 lanes: the table has room for 128 masks
    if (B > 64u) T8[64 + g.lane] = 0ull;
    LdsSpace::sync();
    if (valid) atomicOr(&T8[bk], 1ull << g.lane);
    LdsSpace::sync();
    const uint64_t mates = T8[bk];
    const uint32_t size = (uint32_t)__popcll(mates);
    // members above the lane = size - 1 - members below it: v_mbcnt counts those in two instructions (the mask of the higher lanes
    // takes a 64-bit shift and two ANDs first)
    const uint32_t above = size - 1u - g.below(mates);
    const uint32_t first = valid ? (uint32_t)__builtin_ctzll(mates) : (uint32_t)g.lane;   // (a valid lane's mask holds its own bit: no zero guard)
    const uint32_t lead = (valid && first == (uint32_t)g.lane) ? size : 0u;
    const uint32_t incl = g.template prefix_incl_first<MAXN>(lead);
    const uint32_t run = n_valid - incl;                                   // ranks taken by buckets led from higher lanes (the sizes add up to the valid elements)
    const uint32_t start = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(first << 2), (int)run);
    return start + above;
}

template <int MAXN = 64>
__device__ __forceinline__ void stage_mat_reg(const Work<LdsSpace> &ws, const Grp<64> &g, const uint16_t *OLD, uint16_t *NEW,
                                              uint32_t n_old, uint32_t B, uint32_t M, uint32_t S) {
    const uint32_t t = (uint32_t)g.lane;
    const bool valid = t < B;
    // unpredicated reads (every address lies inside the workspace; what the lanes past the stage read is never used)
    uint32_t pos = t;
    if (n_old) { const uint32_t o = OLD[t]; pos = t < n_old ? o : t; }
    const uint32_t key = ws.D[pos];
    const uint32_t rank = rank_in_registers<MAXN>(g, valid, mod_stage(key, B, M, S), ws.TBL, B, B);
    if (valid) NEW[rank] = (uint16_t)pos;
    LdsSpace::sync();
}

template <int MAXN = 64>
__device__ __forceinline__ Pick stage_final_reg(const Work<LdsSpace> &ws, const Grp<64> &g, const uint16_t *OLD, uint32_t n_old,
                                                uint32_t L, uint32_t B, uint32_t M, uint32_t S, uint32_t rsel) {
    const uint32_t t = (uint32_t)g.lane;
    const bool valid = t < L;
    uint32_t pos = t;
    if (n_old) { const uint32_t o = OLD[t]; pos = t < n_old ? o : t; }
    const uint32_t key = ws.D[pos];
    const uint32_t rank = rank_in_registers<MAXN>(g, valid, mod_stage(key, B, M, S), ws.TBL, B, L);
    // (the valid lanes' mask by scalar arithmetic: a ballot of `valid`, a compare computed blocks earlier, is rebuilt from a 0/1 vector)
    const uint64_t hm = __ballot(rank == rsel) & (~0ull >> (64u - L));                  // 1 <= L <= 64
    const int src = __builtin_ctzll(hm);               // (some lane holds the rank: c >= 1, the mask is never empty)
    return Pick{g.bcast(key, src), g.bcast(pos, src)};
}

// The same for 65..128 elements: lane l holds positions l (slot 0) and 64 + l (slot 1); four mate masks (own slot x other
// slot).  Positions are slot-major, so a bucket's leader is in slot 0 whenever it has a member there, and the buckets led
// from slot 1 (the later positions) are laid out first.
struct Rank2 { uint32_t r0, r1; };
// The four mate masks come through the bucket table: every bucket owns a 128-bit member mask (bit l of word s: the element of
// lane l in slot s), elements set their bit with one 64-bit LDS atomic OR and read their bucket's mask back -- two LDS round
// trips in place of a radix pass of ballots, 12 vector instructions per bit of the bucket number (stage of 127 elements: 208 ->
// 90 vector instructions, 6.05 -> 5.79 ms per 1M walks on C5; for the one-element-per-lane stages the same exchange is neutral
// -- 62-78 -> 37-41 instructions against two more LDS round trips -- and they keep the ballots)
__device__ __forceinline__ void mates2_by_table(uint32_t *TBL, const Grp<64> &g, bool valid1, uint32_t bk0, uint32_t bk1, uint32_t B,
                                                uint64_t &m00, uint64_t &m01, uint64_t &m10, uint64_t &m11) {
    uint4 *T4 = reinterpret_cast<uint4 *>(TBL);
    for (uint32_t i = (uint32_t)g.lane; i < B; i += 64) T4[i] = make_uint4(0u, 0u, 0u, 0u);
    LdsSpace::sync();
    unsigned long long *T8 = reinterpret_cast<unsigned long long *>(TBL);
    const unsigned long long bit = 1ull << g.lane;
    atomicOr(&T8[2u * bk0], bit);
    if (valid1) atomicOr(&T8[2u * bk1 + 1u], bit);
    LdsSpace::sync();
    const uint4 a = T4[bk0], b = T4[bk1];
    m00 = ((uint64_t)a.y << 32) | a.x; m01 = ((uint64_t)a.w << 32) | a.z;
    m10 = ((uint64_t)b.y << 32) | b.x; m11 = ((uint64_t)b.w << 32) | b.z;
}
__device__ __forceinline__ Rank2 rank2_from_mates(const Grp<64> &g, bool valid1, uint64_t m00, uint64_t m01, uint64_t m10, uint64_t m11) {
    const uint64_t gt = ~1ull << g.lane;
    const uint32_t lane = (uint32_t)g.lane;
    // element in slot 0: its bucket's leader is the lowest slot-0 mate (itself included)
    const uint32_t first0 = (uint32_t)__builtin_ctzll(m00);                // (holds the element's own bit)
    const uint32_t size0 = (uint32_t)(__popcll(m00) + __popcll(m01));
    const uint32_t above0 = (uint32_t)(__popcll(m00 & gt) + __popcll(m01));
    const uint32_t lead0 = first0 == lane ? size0 : 0u;
    // element in slot 1: led from slot 0 if it has a mate there, else by the lowest slot-1 mate
    const bool from0 = m10 != 0ull;
    const uint32_t first1 = valid1 ? (uint32_t)__builtin_ctzll(from0 ? m10 : m11) : lane;   // (m11 holds a valid element's own bit)
    const uint32_t above1 = (uint32_t)__popcll(m11 & gt);
    const uint32_t lead1 = (valid1 && !from0 && first1 == lane) ? (uint32_t)__popcll(m11) : 0u;
    // both prefix sums in ONE scan: the sums stay below 2^16 (at most 128 elements), slot 1 rides in the upper half
    const uint32_t incl = g.prefix_incl(lead0 | (lead1 << 16)), tot = g.last(incl);
    const uint32_t incl0 = incl & 0xFFFFu, incl1 = incl >> 16, tot1 = tot >> 16;
    const uint32_t run1 = tot1 - incl1;                                   // buckets led from slot 1, higher lanes first
    const uint32_t run0 = tot1 + ((tot & 0xFFFFu) - incl0);               // then those led from slot 0
    const uint32_t st0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(first0 << 2), (int)run0);
    const uint32_t st1a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(first1 << 2), (int)run0);
    const uint32_t st1b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(first1 << 2), (int)run1);
    Rank2 r;
    r.r0 = st0 + above0;
    r.r1 = (from0 ? st1a : st1b) + above1;
    return r;
}

__device__ __forceinline__ void stage_mat_reg2(const Work<LdsSpace> &ws, const Grp<64> &g, const uint16_t *OLD, uint16_t *NEW,
                                               uint32_t n_old, uint32_t B, uint32_t M, uint32_t S) {
    const uint32_t t0 = (uint32_t)g.lane, t1 = t0 + 64u;
    const bool valid1 = t1 < B;
    uint32_t pos0 = t0, pos1 = t1;
    { const uint32_t o0 = OLD[t0], o1 = OLD[t1]; pos0 = t0 < n_old ? o0 : t0; pos1 = t1 < n_old ? o1 : t1; }
    const uint32_t key0 = ws.D[pos0], key1 = ws.D[pos1];
    uint64_t m00, m01, m10, m11;
    mates2_by_table(ws.TBL, g, valid1, mod_stage(key0, B, M, S), valid1 ? mod_stage(key1, B, M, S) : 0u, B, m00, m01, m10, m11);
    const Rank2 r = rank2_from_mates(g, valid1, m00, m01, m10, m11);
    NEW[r.r0] = (uint16_t)pos0;
    if (valid1) NEW[r.r1] = (uint16_t)pos1;
    LdsSpace::sync();
}

// The LAST stage only has to name the element at iteration position `rsel`, so nothing is ranked or materialised:
// one atomicMin (first position of every bucket) and one atomicAdd (its size) on the same word, a scan of the sizes
// over the bucket leaders to find the bucket that holds position rsel, and ballots among that bucket's few members.
// Nothing in it is predicated per element (measured on C5: 6.57 -> 6.05 ms per 1M walks against `if (valid[j])` around every
// read and a valid flag in every test): reads are unconditional (every address lies inside the walk's workspace), an element
// past the candidates (t >= L) gets bucket number B -- a table slot of its own -- and lanes that hold no candidate at all skip
// the table atomics (all of them on one address would serialise).  So only the lane holding candidate L-1 enters elements past
// L; their bucket's first position is the highest of all, it is laid out FIRST and merely shifts every real rank by its size:
// the search looks for rsel + that size, and no later test asks whether an element is valid.
// PASSES > 1 (a tier whose bucket table holds only 1/PASSES of the final stage's buckets): the table phases run once per bucket range
// -- an element takes part in the pass that owns its bucket -- and the leaders' sizes are collected over the passes; everything
// after the table (scan, search, the one bucket's members) is unchanged.
template <int GS, int NJ, int PASSES = 1>
__device__ __forceinline__ Pick stage_final(const Work<LdsSpace> &ws, const Grp<GS> &g, const uint16_t *OLD, uint32_t n_old,
                                            uint32_t L, uint32_t B, uint32_t M, uint32_t S, uint32_t rsel) {
    const uint32_t t0 = (uint32_t)g.lane * NJ;
    uint32_t pos[NJ], bk[NJ];                                                   // position in D; the key is only needed for the bucket
    const uint32_t H = PASSES > 1 ? (B + (uint32_t)PASSES) / (uint32_t)PASSES : B + 1u;   // buckets (slots 0 .. B) per pass
    if constexpr (PASSES == 1) {
        uint4 *T4 = reinterpret_cast<uint4 *>(ws.TBL);
        const uint32_t n4 = (B + 4u) >> 2;                                      // slots 0 .. B
        for (uint32_t i = g.lane; i < n4; i += GS) T4[i] = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu);
    }
    // unpredicated reads: every address lies inside the walk's workspace, and what the lanes past the candidates read is not used
    {
        uint32_t o[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) o[j] = OLD[t0 + j];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const uint32_t t = t0 + j; pos[j] = t < n_old ? o[j] : t; }
    }
    {
        uint32_t key[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) key[j] = ws.D[pos[j]];
        LdsSpace::sync();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint32_t m = mod_stage(key[j], B, M, S);
            bk[j] = t0 + j < L ? m : B;
        }
    }
    uint32_t gs[NJ], mine_total = 0u;
    if constexpr (PASSES == 1) {
        if (t0 < L) {                                                          // (both atomics under ONE saved mask)
#pragma unroll
            for (int j = 0; j < NJ; ++j) atomicMin(&ws.TBL[bk[j]], t0 + j);
            LdsSpace::sync();
#pragma unroll
            for (int j = 0; j < NJ; ++j) atomicAdd(&ws.TBL[bk[j]], 0x10000u);
        }
        LdsSpace::sync();
#pragma unroll
        for (int j = 0; j < NJ; ++j) gs[j] = ws.TBL[bk[j]];                 // (size << 16) | first position
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) gs[j] = 0xFFFFFFFFu;                   // first position 0xFFFF: no element's own
        for (uint32_t ps = 0; ps < (uint32_t)PASSES; ++ps) {
            const uint32_t b0 = ps * H;
            {
                uint4 *T4 = reinterpret_cast<uint4 *>(ws.TBL);
                const uint32_t n4 = (H + 3u) >> 2;
                for (uint32_t i = g.lane; i < n4; i += GS) T4[i] = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu);
            }
            LdsSpace::sync();
            if (t0 < L) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) if (bk[j] - b0 < H) atomicMin(&ws.TBL[bk[j] - b0], t0 + j);
            }
            LdsSpace::sync();
            if (t0 < L) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) if (bk[j] - b0 < H) atomicAdd(&ws.TBL[bk[j] - b0], 0x10000u);
            }
            LdsSpace::sync();
#pragma unroll
            for (int j = 0; j < NJ; ++j) { const uint32_t lb = bk[j] - b0; const uint32_t v = ws.TBL[lb < H ? lb : 0u]; gs[j] = lb < H ? v : gs[j]; }
            LdsSpace::sync();
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        gs[j] = (gs[j] & 0xFFFFu) == t0 + j ? (gs[j] >> 16) : 0u;       // bucket leaders carry the size
        mine_total += gs[j];
    }
    const uint32_t incl = g.prefix_incl(mine_total);
    const uint32_t past = (uint32_t)NJ - 1u - (L - 1u) % (uint32_t)NJ;    // only the lane that holds candidate L-1 enters its elements past L
    uint32_t run = (L + past) - incl;                                     // positions taken by buckets led from higher lanes (all sizes together: the elements entered)
    const uint32_t target1 = rsel + past + 1u;
    // (no boolean is carried from slot to slot: an OR of lane predicates is a scalar instruction per slot -- the offset's sentinel
    // says afterwards whether the lane holds the bucket)
    uint32_t hb = 0u, ho = 0xFFFFFFFFu;
    {   // all compares first, then the selects: a v_cndmask right behind the v_cmp whose mask it reads costs a two-wait-state filler
        uint32_t e[NJ];
        bool h[NJ];
#pragma unroll
        for (int j = NJ - 1; j >= 0; --j) {
            run += gs[j];
            e[j] = run - target1;                                         // one unsigned compare: run - size <= target < run; e = members of the bucket BELOW the answer
            h[j] = e[j] < gs[j];
        }
#pragma unroll
        for (int j = NJ - 1; j >= 0; --j) { hb = h[j] ? bk[j] : hb; ho = h[j] ? e[j] : ho; }
    }
    const uint64_t hm = g.ballot(ho != 0xFFFFFFFFu);
    // (a bucket holds the target rank and a member of it the offset: the masks are never empty; unguarded, the lane number is the
    // find-first-set alone -- the `mask ? .. : 0` form costs a scalar compare and select each)
    const int hsrc = GS == 64 ? __builtin_ctzll(hm) : (hm ? (__ffsll((long long)hm) - 1) : 0);
    const uint32_t bstar = g.bcast(hb, hsrc), off = g.bcast(ho, hsrc);
    // members of that bucket are visited in DESCENDING position: the answer has exactly `off` members BELOW it (counted from below,
    // the prefix sum needs no total).  Members in lower lanes: the lane's own member count, summed over the lanes by ONE scan (not
    // one ballot per element slot)
    uint32_t cj[NJ], below = 0u;
    if constexpr (GS == 64) {
        // one walk per wave: the members in lower lanes are counted from the slots' ballots -- two v_mbcnt per slot, each adding to
        // the running count -- where a DPP scan of the lanes' own counts took six steps with a wait-state filler behind each
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            cj[j] = bk[j] == bstar ? 1u : 0u;
            const uint64_t mj = __ballot(bk[j] == bstar);
            below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mj >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mj, below));
        }
    } else {
        uint32_t own = 0u;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { cj[j] = bk[j] == bstar ? 1u : 0u; own += cj[j]; }
        below = g.prefix_incl(own) - own;
    }
    uint32_t mine = 0xFFFFFFFFu;
    {
        bool hit[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint32_t bl = cj[j] != 0u ? below : 0xFFFFFFFFu;       // a member of the bucket with exactly `off` members below it
            hit[j] = bl == off;
            below += cj[j];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) mine = hit[j] ? pos[j] : mine;
    }
    const uint64_t mk = g.ballot(mine != 0xFFFFFFFFu);
    const int src = GS == 64 ? __builtin_ctzll(mk) : (mk ? (__ffsll((long long)mk) - 1) : 0);
    LdsSpace::sync();
    const uint32_t q = g.bcast(mine, src);
    return Pick{g.uni(ws.D[q]), q};
}

// LDS tiers: iteration-order selection with the register-resident stages; `nvalid` = number of leading stages whose
// materialised order is still valid (the candidates they cover did not change since they were computed).
// An LDS tier only ever touches the first NST+1 entries of the bucket chain, so the stage index is a template parameter:
// B, the multiply-shift constants and the order-array offsets are immediates (no scalar loads from the chain table, no
// waits on them), the elements-per-lane variant of a materialising stage is fixed at compile time, and a final stage only
// carries the variants its candidate range (B[stage-1], min(B[stage], CAP)] can need.
constexpr int nj_of(int per) { return per <= 1 ? 1 : per <= 3 ? 3 : per <= 5 ? 5 : per <= 7 ? 7 : per <= 9 ? 9 : per <= 13 ? 13 : per <= 17 ? 17 : per <= 19 ? 19 : 33; }
constexpr int nst_of(int cap) { return cap <= 32 ? 2 : (cap <= 64 ? 3 : (cap <= 512 ? 5 : (cap <= 1024 ? 6 : 7))); }       // stages that are ever materialised
// Bucket-table words of a tier.  Normally the smallest chain value >= CAP (the final stage's buckets in one piece).  The 704-candidate
// tier keeps only HALF of its final stage's 1109 + 1 slots (stage_final runs two passes there) and lets the 541-bucket stage put its
// position lists into its own order array: 11.6 KB per walk instead of 19.4 KB, 12 walks per CU instead of 8.
// The 1408-candidate tier likewise keeps the 1109 words its largest materialising stage needs and runs its final stage (2357 + 1
// slots) in three passes: 23.2 KB per walk instead of 38.9 KB, 6 walks per CU instead of 4.
constexpr int tier_index(int cap) { return cap <= 64 ? 0 : (cap <= 512 ? 1 : (cap <= 704 ? 2 : (cap <= 1024 ? 3 : (cap <= 1408 ? 4 : 5)))); }
// (The same diet for the 1024-candidate tier -- 17.2 KB, 9 walks per CU instead of 8 -- measured slower: 50.9 against 52.0 M/s on
// degree 160, k = 6: one more resident walk does not pay for the second pass of every final stage above 541 candidates.)
// (The 32-candidate form of the 8-lane tier -- batches of graphs of at most 33 vertices, QM9- and MUTAG-sized -- ends at the 59-bucket stage.)
constexpr int tbl_words(int cap) { constexpr int w[6] = {127, 541, 555, 1109, 1109, 2357}; return ((cap <= 32 ? 59 : w[tier_index(cap)]) + 3) & ~3; }

template <int STAGE> struct ChainAt {
    static constexpr uint32_t B = kChainHost[STAGE], M = cmagic(kChainHost[STAGE]), S = (uint32_t)clog2(kChainHost[STAGE]) - 1u;
    static constexpr uint32_t O = ord_words_before(STAGE);
    static constexpr uint32_t NOLD = STAGE ? kChainHost[STAGE ? STAGE - 1 : 0] : 0u;
    static constexpr uint32_t OOLD = STAGE ? ord_words_before(STAGE ? STAGE - 1 : 0) : 0u;
};

template <int GS, int MAXPER, int STAGE>
__device__ __forceinline__ void mat_at(const Work<LdsSpace> &ws, const Grp<GS> &g) {
    using C = ChainAt<STAGE>;
    const uint16_t *OLD = ws.ORD + C::OOLD;                                   // stage 0 has no predecessor (NOLD = 0: never read)
    uint16_t *NEW = ws.ORD + C::O;
    constexpr int per = (int)((C::B + GS - 1) / GS);                          // these stages are full: L == B
    STAMP_SUB_BEGIN();
    // position lists behind the table only where the tier's table has the room (see tbl_words)
    constexpr bool POS_IN_NEW = (int)(((C::B + 3u) & ~3u) + (C::B + 1u) / 2u) > tbl_words(MAXPER * GS);
    if constexpr (GS == 64 && per <= 1) stage_mat_reg<(int)C::B>(ws, g, OLD, NEW, C::NOLD, C::B, C::M, C::S);
    // (for 65..128 elements the register ranking, 203 VALU instructions, still beats the bucket-table variant with 2 elements per
    // lane, 169 instructions but nine LDS round trips: 6.81 against 6.95 ms per 1M walks)
    else if constexpr (GS == 64 && per <= 2) stage_mat_reg2(ws, g, OLD, NEW, C::NOLD, C::B, C::M, C::S);
    else stage_mat<GS, nj_of(per), POS_IN_NEW>(ws, g, OLD, NEW, C::NOLD, C::B, C::M, C::S);
    STAMP_SUB_END(per <= 1 ? 10 : (per <= 2 ? 11 : 12));
}

// The variants of a final stage by elements per lane: variant i serves per <= var_hi(i) (and more than var_hi(i-1)) with var_nj(i)
// element slots per lane (0: the register-ranked final of one walk per wave).  Which variant serves how many elements per lane
// (measured on C5, VALU-bound at 20 waves/CU): up to 64 candidates the register-ranked final (78 VALU instructions against 59 + 7
// LDS operations for the table variant: the shorter dependency chain wins); from 65 on the bucket-table variant with exactly
// ceil(c/64) elements per lane -- for 65..128 candidates it takes 93 VALU instructions where ranking two elements per lane in
// registers took 207 (7.16 -> 7.01 ms per 1M walks).
template <int GS> constexpr int var_count() { return GS == 64 ? 13 : 10; }
template <int GS> constexpr int var_hi(int i) {
    constexpr int h64[13] = {1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 17, 22, 1 << 20}, h8[10] = {1, 3, 5, 7, 9, 11, 13, 17, 22, 1 << 20};
    return GS == 64 ? h64[i] : h8[i];
}
template <int GS> constexpr int var_nj(int i) {
    constexpr int n64[13] = {0, 2, 3, 4, 5, 6, 7, 9, 11, 13, 17, 22, 33}, n8[10] = {1, 3, 5, 7, 9, 11, 13, 17, 22, 33};
    return GS == 64 ? n64[i] : n8[i];
}
template <int GS> constexpr int var_first_with_hi_at_least(int per) { int i = 0; while (var_hi<GS>(i) < per) ++i; return i; }

// the variant by a binary decision tree over the variant indices [I0, I1] the stage's candidate range can need: every leaf returns,
// nothing follows the tree.  (A sequence of `if (per <= HI) return variant;` cases compiles into a chain in which every case BEHIND the
// one taken is skipped by a flag test and a taken branch -- three scalar instructions and 18 wave-cycles of branch per skipped case.)
template <int GS, int MAXPER, int STAGE, int I0, int I1>
__device__ __forceinline__ Pick final_tree(const Work<LdsSpace> &ws, const Grp<GS> &g, int per, uint32_t c, uint32_t rsel) {
    using C = ChainAt<STAGE>;
    if constexpr (I0 == I1) {
        const uint16_t *OLD = ws.ORD + C::OOLD;
        constexpr uint32_t CAP = (uint32_t)(MAXPER * GS);
        constexpr int NJ = var_nj<GS>(I0);
        // the table holds the stage's slots 0 .. B in one piece, or a half / a third of them
        constexpr int PASSES = ((int)C::B + tbl_words((int)CAP)) / tbl_words((int)CAP);
        static_assert(((int)C::B + PASSES) / PASSES <= tbl_words((int)CAP), "a pass of the final stage must fit the tier's bucket table");
        STAMP_SUB_BEGIN();
        Pick r;
        if constexpr (NJ == 0) r = stage_final_reg<(int)(C::B < 64u ? C::B : 64u)>(ws, g, OLD, C::NOLD, c, C::B, C::M, C::S, rsel);
        else r = stage_final<GS, NJ, PASSES>(ws, g, OLD, C::NOLD, c, C::B, C::M, C::S, rsel);
        STAMP_SUB_END(var_hi<GS>(I0) <= 1 ? 13 : (var_hi<GS>(I0) <= 2 ? 14 : 15));
        return r;
    } else {
        constexpr int MID = (I0 + I1) / 2;
        if (per <= var_hi<GS>(MID)) return final_tree<GS, MAXPER, STAGE, I0, MID>(ws, g, per, c, rsel);
        return final_tree<GS, MAXPER, STAGE, MID + 1, I1>(ws, g, per, c, rsel);
    }
}

template <int GS, int MAXPER, int STAGE>
__device__ __forceinline__ Pick final_at(const Work<LdsSpace> &ws, const Grp<GS> &g, uint32_t c, uint32_t rsel) {
    using C = ChainAt<STAGE>;
    constexpr uint32_t CAP = (uint32_t)(MAXPER * GS);
    // a final at this stage sees NOLD < c <= min(B, CAP) candidates: only the variants that range can need are instantiated
    constexpr int pmin = (int)((C::NOLD + 1u + GS - 1) / GS), pmax = (int)(((C::B < CAP ? C::B : CAP) + GS - 1) / GS);
    const int per = (int)((c + GS - 1) / GS);
    return final_tree<GS, MAXPER, STAGE, var_first_with_hi_at_least<GS>(pmin), var_first_with_hi_at_least<GS>(pmax)>(ws, g, per, c, rsel);
}

// the stages to compute come as a bit set: one scalar bit test and branch per stage
template <int GS, int MAXPER, int STAGE, int NST>
__device__ __forceinline__ void materialise_bits(const Work<LdsSpace> &ws, const Grp<GS> &g, uint32_t need) {
    if constexpr (STAGE < NST) {
        if (need & (1u << STAGE)) mat_at<GS, MAXPER, STAGE>(ws, g);
        materialise_bits<GS, MAXPER, STAGE + 1, NST>(ws, g, need);
    }
}

// the final stage's variant by a binary decision tree over the stage index [LO, HI] (a chain of `fs == STAGE` tests costs the late,
// expensive steps four taken branches before their final starts; a taken branch is 18 wave-cycles)
template <int GS, int MAXPER, int LO, int HI>
__device__ __forceinline__ Pick final_from(const Work<LdsSpace> &ws, const Grp<GS> &g, int fs, uint32_t c, uint32_t rsel) {
    if constexpr (LO == HI) {
        return final_at<GS, MAXPER, LO>(ws, g, c, rsel);
    } else {
        constexpr int MID = (LO + HI + 1) / 2;
        if (fs >= MID) return final_from<GS, MAXPER, MID, HI>(ws, g, fs, c, rsel);
        return final_from<GS, MAXPER, LO, MID - 1>(ws, g, fs, c, rsel);
    }
}

// One walk per wave: lane i < NST keeps B[i] in a register for the whole kernel (the other lanes 0xFFFFFFFF), and the number of chain
// values below x is the popcount of ONE compare's ballot -- a vector and a scalar instruction where the sign-bit sum below takes
// 2 NST + (NST - 1) scalar ones.  The scalar unit is the walk kernel's busiest issue port (tools/pad_probe.sh: a scalar instruction
// costs a wave 12 cycles at the margin, a 4-byte vector one 7), and this runs twice per growth step.
template <int NST> __device__ __forceinline__ uint32_t chain_lane_const(int lane) {
    uint32_t v = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < NST; ++i) v = lane == i ? kChainHost[i] : v;
    asm volatile("" : "+v"(v));                                             // keep it: do not rebuild it at every use
    return v;
}
__device__ __forceinline__ int chain_index_below_lanes(uint32_t x, uint32_t chain_lane) {
    int n = (int)__popcll(__ballot(chain_lane < x));
    // opaque 32-bit value: the compiler otherwise compares the 64-bit population count itself, and an ordered 64-bit compare of
    // scalars has no scalar instruction (v_cmp_lt_u64 + s_and + s_cbranch_vccz per test of the stage dispatch)
    asm volatile("" : "+s"(n));
    return n;
}
template <int NST> __device__ __forceinline__ int chain_index_below(uint32_t x) {     // number of the first NST chain values < x
    // x <= 2^16: the sign bit of B[i] - x says x > B[i] -- two scalar instructions per term and no condition codes (the
    // compare-and-add form went through a lane mask and a VGPR per term: 6.77 -> 6.61 ms per 1M walks together with the
    // bit-set dispatch of materialise_bits)
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < NST; ++i) n += (kChainHost[i] - x) >> 31;
    return (int)n;
}

template <int GS, int MAXPER>
__device__ __forceinline__ Pick select_lds(const Work<LdsSpace> &ws, const Grp<GS> &g_, uint32_t c, uint32_t rsel, int &nvalid) {
    constexpr int NST = nst_of(MAXPER * GS);
    // The stages address LDS by lane * (elements per lane) and compare the lane with constants; the compiler hoists all of that
    // out of the walk loops, runs out of registers and reloads it from scratch memory right where a stage starts (a memory
    // round trip in front of its first LDS read).  An opaque copy of the lane index keeps those few instructions inside
    // (C5: 96 VGPRs + 52 bytes of scratch -> 88 VGPRs, none; 5.68 -> 5.53 ms per 1M walks).
    Grp<GS> g = g_;
    asm volatile("" : "+v"(g.lane));
    int fs;
    if constexpr (GS == 64) fs = chain_index_below_lanes(c, g_.chain); else fs = chain_index_below<NST>(c);
    const uint32_t need = ((1u << fs) - 1u) & ~((1u << nvalid) - 1u);        // stages nvalid .. fs-1
    if (need) materialise_bits<GS, MAXPER, 0, NST>(ws, g, need);             // (nothing to bring up to date in a third of the steps: one test instead of NST)
    nvalid = nvalid > fs ? nvalid : fs;
    return final_from<GS, MAXPER, 0, NST>(ws, g, fs, c, rsel);
}

template <int GS, int MAXPER> __device__ __forceinline__ Pick select_any(const Work<LdsSpace> &ws, const Grp<GS> &g, uint32_t c, uint32_t rsel, int &nvalid) {
    return select_lds<GS, MAXPER>(ws, g, c, rsel, nvalid);
}
template <int GS, int MAXPER> __device__ __forceinline__ Pick select_any(const Work<GlbSpace> &ws, const Grp<GS> &g, uint32_t c, uint32_t rsel, int &nvalid) {
    return Pick{select_in_order<GS, GlbSpace>(ws, g, c, rsel, nvalid), c};
}

// Adjacency row of the vertex just added to the sample (local index size-1):
//   - counts the induced-edge entries it contributes (entries to earlier members count twice: the symmetric CSR holds
//     the mirror entry in the earlier member's row; entries to itself count once),
//   - ADD: appends neighbours that pass the suffix filter and were never seen to D (first-occurrence order).
// Returns false when the walk outgrew this tier's workspace.
// key word of the membership hash: vertex id (< 2^30) | kFresh (inserted by the chunk being processed) | kInS (sampled)
constexpr uint32_t kKeyMask = 0x3FFFFFFFu, kFresh = 0x40000000u, kInS = 0x80000000u;

// Edge staging (one walk per wave, LDS tiers): every entry of the scanned row that points into the sample IS an induced
// edge between the new vertex (local index size-1) and an earlier member; its CSR position and the two local indices are
// kept in a short LDS list and turned into the row's output items at the end of the walk (stage_flush).
struct StageCtx {
    uint4 *EL;               // [UGS_STAGE_ENTRIES]: x = CSR position, y = the member the entry points to, z = scanned vertex's local index
    uint32_t ne;             // hits so far (may exceed the list: the row then goes to the row-reading fill kernel)
    bool on;
    uint64_t onm;            // all ones when staging is on: `hits & onm` tests "on, and any hit" with one scalar AND
};

template <int GS>
__device__ __forceinline__ void stage_hits(StageCtx &sc, const Grp<GS> &g, bool in_s, uint32_t w, uint32_t p, uint32_t size) {
    const uint64_t im = g.ballot(in_s);
    if (!im) return;
    const uint32_t slot = sc.ne + g.below(im);
    if (in_s && slot < UGS_STAGE_ENTRIES) sc.EL[slot] = make_uint4(p, w, size - 1u, 0u);
    sc.ne += (uint32_t)__popcll(im);
}

// The same with the hits' lane mask taken by the caller in the block that computes the compare (one walk per wave).  A ballot asked
// for in a LATER block than its compare is rebuilt by the compiler from a 0/1 vector (v_cndmask + v_cmp_ne and a wait-state filler
// between them); a mask taken on the spot is the compare's own result, stays in scalar registers, and comes back as a lane
// predicate for free (inverse ballot: the mask goes straight into EXEC).
__device__ __forceinline__ void stage_hits_mask(StageCtx &sc, const Grp<64> &g, uint64_t im, uint32_t w, uint32_t p, uint32_t size) {
    im &= sc.onm;
    if (!im) return;
    const uint32_t slot = sc.ne + g.below(im);
    if (__builtin_amdgcn_inverse_ballot_w64(im) && slot < UGS_STAGE_ENTRIES) sc.EL[slot] = make_uint4(p, w, size - 1u, 0u);
    sc.ne += (uint32_t)__popcll(im);
}

// The probe loops of the membership table for the one-walk-per-wave LDS tiers, written against the EXEC mask directly.  Every
// active lane probes its own key; a lane is done when it meets an empty slot (and, inserting, has taken it) or its key.  The
// compiler's version of this loop -- unrolled eight times, one saved EXEC mask per level and two scalar instructions per level on
// the way out -- spends ~6 scalar instructions per probe; here the two `v_cmpx` narrow EXEC to the lanes still probing, the loop
// ends when EXEC is empty and the entry mask is restored once.  No scalar ALU instruction inside the loop.
// The table is never full (scan_chunk's hlimit guard), and the probe step is odd, so every lane terminates.
// The loops start by narrowing EXEC to the CANDIDATES (entries whose rank passes the suffix filter: one v_cmpx) -- an `if (cand)` around
// the asm made the compiler build an if/else of saved masks (five scalar instructions and two duplicated vector ones per chunk).
__device__ __forceinline__ uint32_t probe_insert_lds(uint32_t *HK, uint32_t hmask, uint32_t slot, uint32_t step, uint32_t w, uint32_t &seen,
                                                     int rank, uint32_t root_vi) {
    const uint32_t base = (uint32_t)(uintptr_t)HK;                            // low half of a flat LDS pointer = its LDS address
    uint32_t addr = base + (slot << 2), t;
    uint64_t saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "v_cmpx_le_i32 %[rv], %[rk]\n"
        "ugs_pi_%=:\n"
        "ds_cmpst_rtn_b32 %[seen], %[addr], %[empty], %[neww]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_and_b32 %[t], 0x3fffffff, %[seen]\n"
        "v_cmpx_ne_u32 -1, %[seen]\n"
        "v_cmpx_ne_u32 %[t], %[w]\n"
        "s_cbranch_execz ugs_pd_%=\n"
        "v_add_u32 %[slot], %[slot], %[step]\n"
        "v_and_b32 %[slot], %[hmask], %[slot]\n"
        "v_lshl_add_u32 %[addr], %[slot], 2, %[base]\n"
        "s_branch ugs_pi_%=\n"
        "ugs_pd_%=:\n"
        "s_mov_b64 exec, %[sv]\n"
        : [seen] "+v"(seen), [slot] "+v"(slot), [addr] "+v"(addr), [t] "=&v"(t), [sv] "=&s"(saved)
        : [empty] "v"(kEmpty), [neww] "v"(w | kFresh), [w] "v"(w), [step] "v"(step), [hmask] "s"(hmask), [base] "s"(base), [rk] "v"(rank), [rv] "s"(root_vi)
        : "vcc", "memory");
    return slot;
}
__device__ __forceinline__ void probe_find_lds(const uint32_t *HK, uint32_t hmask, uint32_t slot, uint32_t step, uint32_t w, uint32_t &seen,
                                               int rank, uint32_t root_vi) {
    const uint32_t base = (uint32_t)(uintptr_t)HK;
    uint32_t addr = base + (slot << 2), t;
    uint64_t saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "v_cmpx_le_i32 %[rv], %[rk]\n"
        "ugs_pf_%=:\n"
        "ds_read_b32 %[seen], %[addr]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_and_b32 %[t], 0x3fffffff, %[seen]\n"
        "v_cmpx_ne_u32 -1, %[seen]\n"
        "v_cmpx_ne_u32 %[t], %[w]\n"
        "s_cbranch_execz ugs_pe_%=\n"
        "v_add_u32 %[slot], %[slot], %[step]\n"
        "v_and_b32 %[slot], %[hmask], %[slot]\n"
        "v_lshl_add_u32 %[addr], %[slot], 2, %[base]\n"
        "s_branch ugs_pf_%=\n"
        "ugs_pe_%=:\n"
        "s_mov_b64 exec, %[sv]\n"
        : [seen] "+v"(seen), [slot] "+v"(slot), [addr] "+v"(addr), [t] "=&v"(t), [sv] "=&s"(saved)
        : [w] "v"(w), [step] "v"(step), [hmask] "s"(hmask), [base] "s"(base), [rk] "v"(rank), [rv] "s"(root_vi)
        : "vcc", "memory");
}

// The same loops for the tiers with several walks per wave (8 or 16 lanes per walk): the table's address and the root's rank differ
// between the wave's groups, so they come in vector registers.  (The compiler's versions of these divergent loops -- a compare-and-swap
// with two exits -- rebuild the EXEC mask with a dozen scalar instructions per trip.)
__device__ __forceinline__ uint32_t probe_insert_lds_v(uint32_t *HK, uint32_t hmask, uint32_t slot, uint32_t step, uint32_t w, uint32_t &seen,
                                                       int rank, uint32_t root_vi) {
    const uint32_t base = (uint32_t)(uintptr_t)HK;
    uint32_t addr = base + (slot << 2), t;
    uint64_t saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "v_cmpx_le_i32 %[rv], %[rk]\n"
        "ugs_qi_%=:\n"
        "ds_cmpst_rtn_b32 %[seen], %[addr], %[empty], %[neww]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_and_b32 %[t], 0x3fffffff, %[seen]\n"
        "v_cmpx_ne_u32 -1, %[seen]\n"
        "v_cmpx_ne_u32 %[t], %[w]\n"
        "s_cbranch_execz ugs_qd_%=\n"
        "v_add_u32 %[slot], %[slot], %[step]\n"
        "v_and_b32 %[slot], %[hmask], %[slot]\n"
        "v_lshl_add_u32 %[addr], %[slot], 2, %[base]\n"
        "s_branch ugs_qi_%=\n"
        "ugs_qd_%=:\n"
        "s_mov_b64 exec, %[sv]\n"
        : [seen] "+v"(seen), [slot] "+v"(slot), [addr] "+v"(addr), [t] "=&v"(t), [sv] "=&s"(saved)
        : [empty] "v"(kEmpty), [neww] "v"(w | kFresh), [w] "v"(w), [step] "v"(step), [hmask] "s"(hmask), [base] "v"(base), [rk] "v"(rank), [rv] "v"(root_vi)
        : "vcc", "memory");
    return slot;
}
__device__ __forceinline__ void probe_find_lds_v(const uint32_t *HK, uint32_t hmask, uint32_t slot, uint32_t step, uint32_t w, uint32_t &seen,
                                                 int rank, uint32_t root_vi) {
    const uint32_t base = (uint32_t)(uintptr_t)HK;
    uint32_t addr = base + (slot << 2), t;
    uint64_t saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "v_cmpx_le_i32 %[rv], %[rk]\n"
        "ugs_qf_%=:\n"
        "ds_read_b32 %[seen], %[addr]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_and_b32 %[t], 0x3fffffff, %[seen]\n"
        "v_cmpx_ne_u32 -1, %[seen]\n"
        "v_cmpx_ne_u32 %[t], %[w]\n"
        "s_cbranch_execz ugs_qe_%=\n"
        "v_add_u32 %[slot], %[slot], %[step]\n"
        "v_and_b32 %[slot], %[hmask], %[slot]\n"
        "v_lshl_add_u32 %[addr], %[slot], 2, %[base]\n"
        "s_branch ugs_qf_%=\n"
        "ugs_qe_%=:\n"
        "s_mov_b64 exec, %[sv]\n"
        : [seen] "+v"(seen), [slot] "+v"(slot), [addr] "+v"(addr), [t] "=&v"(t), [sv] "=&s"(saved)
        : [w] "v"(w), [step] "v"(step), [hmask] "s"(hmask), [base] "v"(base), [rk] "v"(rank), [rv] "v"(root_vi)
        : "vcc", "memory");
}

// One chunk of an adjacency row: lane holds entry e (neighbour, rank) at CSR position p (a plan has < 2^31 entries); lanes
// without an entry hold kNoEntry, whose rank -1 fails the suffix filter, so `cand` is one signed compare (ranks are < 2^30).
// (their vertex number, all ones, equals no vertex: a test `w == some candidate` needs no `cand &&` in front)
#define UGS_NO_ENTRY make_int2(-1, -1)
// GUARD = false: the caller has checked that the whole row fits (candidates and table entries), so the per-chunk tests are left out.
template <int GS, class SP, bool ADD, bool STG, bool GUARD = true>
__device__ __forceinline__ bool scan_chunk(const Work<SP> &ws, const Grp<GS> &g, uint32_t v, uint32_t root_vi, uint32_t size, uint32_t &c,
                                           uint32_t &hcount, uint32_t &ecount, StageCtx &sc, int2 e, uint32_t p) {
    const uint32_t w = (uint32_t)e.x;
    const bool cand = e.y >= (int)root_vi;
    uint32_t slot = hash_slot(w, ws.hmask);
    const uint32_t step = hash_step(w);
    bool in_s = false;
    if constexpr (GS == 64 && sizeof(typename SP::TW) == 4) {
        // One walk per wave, LDS workspace: every lane predicate is taken as a lane MASK in the block that computes it and carried
        // in scalar registers (see stage_hits_mask); the probes narrow EXEC to the candidates themselves.
        if (ADD) {
            if constexpr (GUARD) { if (hcount + (uint32_t)__popcll(__ballot(cand)) > ws.hlimit) return false; }
            uint32_t seen = kKeyMask;                             // what a lane that is no candidate keeps: not empty, equal to no vertex
            STAMP_SUB_BEGIN();
            slot = probe_insert_lds(ws.HK, ws.hmask, slot, step, w, seen, e.y, root_vi);
            STAMP_SUB_END_OF(1, 6);
            const uint64_t insm = __ballot(seen == kEmpty);                                     // inserted by this lane
            const uint64_t im = __ballot((seen & (kKeyMask | kInS)) == (w | kInS));             // found, and a member of the sample
            uint64_t dupm = __ballot((seen & (kKeyMask | kFresh)) == (w | kFresh));             // found, and inserted by this very chunk
            SP::sync();
            // first occurrences (see below): the inserting lanes, except that a vertex repeated inside the chunk is represented by
            // the LOWEST lane holding it
            uint64_t fm = insm;
            while (dupm) {
                const uint32_t wi = g.bcast(w, __builtin_ctzll(dupm));
                const uint64_t grp = __ballot(w == wi);           // (a vertex has one rank: its lanes are candidates all or none)
                fm = (fm & ~grp) | (grp & (0ull - grp));
                dupm &= ~grp;
            }
            if (__builtin_amdgcn_inverse_ballot_w64(insm)) ws.HK[slot] = w;                      // the chunk is over for this key: drop kFresh
            ecount += __builtin_amdgcn_inverse_ballot_w64(im) ? (w == v ? 1u : 2u) : 0u;         // per lane, summed over the wave at the end of the walk
            if constexpr (STG) stage_hits_mask(sc, g, im, w, p, size);
            const uint32_t nnew = (uint32_t)__popcll(fm);
            if constexpr (GUARD) { if (c + nnew > ws.cap) return false; }
            if (__builtin_amdgcn_inverse_ballot_w64(fm)) ws.D[c + g.below(fm)] = w;
            c += nnew;
            hcount += nnew;
            SP::sync();
        } else {
            uint32_t seen = kEmpty;
            probe_find_lds(ws.HK, ws.hmask, slot, step, w, seen, e.y, root_vi);
            const uint64_t im = __ballot((seen & (kKeyMask | kInS)) == (w | kInS));             // kEmpty (no candidate, or not seen) matches no vertex
            ecount += __builtin_amdgcn_inverse_ballot_w64(im) ? (w == v ? 1u : 2u) : 0u;
            if constexpr (STG) stage_hits_mask(sc, g, im, w, p, size);
        }
        return true;
    } else
    if (ADD) {
        if constexpr (GUARD) { if (hcount + (uint32_t)__popcll(g.ballot(cand)) > ws.hlimit) return false; }
        // probe: only `seen` and `slot` are carried round the loop; what happened is read off `seen` afterwards
        uint32_t seen = kKeyMask;                                 // a value no probe returns for a candidate
        STAMP_SUB_BEGIN();
        if constexpr (sizeof(typename SP::TW) == 4) {
            slot = probe_insert_lds_v(reinterpret_cast<uint32_t *>(ws.HK), ws.hmask, slot, step, w, seen, e.y, root_vi);
        } else
        if (cand) {
            for (uint32_t it = 0; it <= ws.hmask; ++it) {         // the table is never full
                seen = atomicCAS(&ws.HK[slot], kEmpty, w | kFresh);
                if (seen == kEmpty || (seen & kKeyMask) == w) break;
                slot = (slot + step) & ws.hmask;
            }
        }
        STAMP_SUB_END_OF(1, 6);
        // `seen` of a lane that is no candidate keeps kKeyMask: not empty, its key part (all ones) equals no vertex, so every
        // predicate below is a single compare whose ballot the compiler takes straight from v_cmp
        // (the key part of kEmpty is all ones too), and the flags are tested together with the key
        const bool inserted = seen == kEmpty;
        in_s = (seen & (kKeyMask | kInS)) == (w | kInS);                     // found, and a member of the sample
        const bool fresh_dup = (seen & (kKeyMask | kFresh)) == (w | kFresh);  // found, and inserted by this very chunk
        SP::sync();
        // A vertex is NEW if this chunk inserted it; its place in D is that of its FIRST occurrence in the row.  Lanes
        // that met a key inserted by this very chunk (a repeated neighbour, e.g. both directions of a PyG edge) are
        // resolved per repeated vertex: the lowest lane holding it represents it.
        bool first = inserted;
        uint64_t dupm = g.ballot(fresh_dup);
        while (dupm) {
            const int li = __ffsll((long long)dupm) - 1;
            const uint32_t wi = g.bcast(w, li);
            const bool mine = cand && w == wi;
            const uint64_t grp = g.ballot(mine);
            if (mine) first = g.lane == (__ffsll((long long)grp) - 1);
            dupm &= ~grp;
        }
        if (inserted) ws.HK[slot] = w;                            // the chunk is over for this key: drop kFresh
        if constexpr (GS == 64) ecount += in_s ? (w == v ? 1u : 2u) : 0u;       // per lane, summed over the wave at the end of the walk
        else
        {   // entries to earlier members count twice (the mirror entry), entries to the scanned vertex itself once
            const uint64_t im = g.ballot(in_s), sm = g.ballot(w == v);
            ecount += 2u * (uint32_t)__popcll(im & ~sm) + (uint32_t)__popcll(im & sm);
        }
        if constexpr (STG) { if (sc.on) stage_hits<GS>(sc, g, in_s, w, p, size); }
        const uint64_t fm = g.ballot(first);
        const uint32_t nnew = (uint32_t)__popcll(fm);
        if constexpr (GUARD) { if (c + nnew > ws.cap) return false; }
        if (first) ws.D[c + g.below(fm)] = w;
        c += nnew;
        hcount += nnew;
        SP::sync();
    } else {
        uint32_t seen = kEmpty;
        if constexpr (sizeof(typename SP::TW) == 4) {
            probe_find_lds_v(reinterpret_cast<const uint32_t *>(ws.HK), ws.hmask, slot, step, w, seen, e.y, root_vi);
        } else
        if (cand) {
            for (uint32_t it = 0; it <= ws.hmask; ++it) {
                seen = ws.HK[slot];
                if (seen == kEmpty || (seen & kKeyMask) == w) break;
                slot = (slot + step) & ws.hmask;
            }
        }
        in_s = (seen & (kKeyMask | kInS)) == (w | kInS);                     // kEmpty (no candidate, or not seen) matches no vertex
        if constexpr (GS == 64) ecount += in_s ? (w == v ? 1u : 2u) : 0u;
        else
        {
            const uint64_t im = g.ballot(in_s), sm = g.ballot(w == v);
            ecount += 2u * (uint32_t)__popcll(im & ~sm) + (uint32_t)__popcll(im & sm);
        }
        if constexpr (STG) { if (sc.on) stage_hits<GS>(sc, g, in_s, w, p, size); }
    }
    return true;
}

// row of v through the row pointer (8-lane tier, global-memory tier, plans without padded rows)
template <int GS, class SP, bool ADD, bool STG>
__device__ __forceinline__ bool scan_row(const Work<SP> &ws, const Grp<GS> &g, const int2 *adj, uint32_t v,
                                         uint32_t root_vi, uint32_t size, uint32_t &c, uint32_t &hcount,
                                         uint32_t &ecount, uint32_t r0, uint32_t r1, StageCtx &sc) {
    for (uint32_t base = r0; base < r1; base += GS) {
        const uint32_t p = base + (uint32_t)g.lane;
        int2 e = UGS_NO_ENTRY;
        if (p < r1) e = adj[p];
        if (!scan_chunk<GS, SP, ADD, STG>(ws, g, v, root_vi, size, c, hcount, ecount, sc, e, p)) return false;
    }
    return true;
}

// padded row of v (one walk per wave): `e0` = entry `lane` of the row's block, loaded by the caller as soon as v was known.
// Lane 0 holds the header (degree, CSR position of the first entry), lanes 1.. the first entries; a longer row continues in adj[].
__device__ __forceinline__ int2 prow_entry(const UgsPlanDev &P, int64_t vrow, int lane) {
    // the padded rows of a plan span less than 4 GB (ensure_prow), so an entry's byte offset fits 32 bits: one shift-add per lane
    // and the scalar base in the load, instead of 64-bit vector address arithmetic
    const uint32_t off = (((uint32_t)vrow << P.prow_shift) + (uint32_t)lane) << 3;
    return *reinterpret_cast<const int2 *>(reinterpret_cast<const char *>(P.prow) + off);
}
__device__ __forceinline__ int2 load_prow(const UgsPlanDev &P, int64_t vrow, int lane) {
    int2 e = make_int2(0, 0);
    if (lane < P.prow_first) e = prow_entry(P, vrow, lane);
    return e;
}

template <class SP, bool ADD, bool STG>
__device__ __forceinline__ bool scan_prow(const Work<SP> &ws, const Grp<64> &g, const UgsPlanDev &P, uint32_t v,
                                          uint32_t root_vi, uint32_t size, uint32_t &c, uint32_t &hcount,
                                          uint32_t &ecount, int2 e0, int64_t vrow, StageCtx &sc) {
    const uint32_t deg = g.bcast((uint32_t)e0.x, 0);
    const uint32_t start = g.bcast((uint32_t)e0.y, 0);
    const uint32_t inl = (1u << P.prow_shift) - 1u;                              // entries held by the block itself
    const uint32_t n0 = deg < inl ? deg : inl;
    if (n0 >= (uint32_t)P.prow_first) {                                          // the row reaches into the lines not fetched yet
        if (g.lane >= P.prow_first && g.lane <= (int)n0) e0 = prow_entry(P, vrow, g.lane);
    }
    if ((uint32_t)(g.lane - 1) >= n0) e0 = UGS_NO_ENTRY;                        // the header's lane and the lanes behind the row
    // a row whose every entry could be a new candidate and still fit needs no per-chunk overflow tests (nearly all rows); the
    // others take the guarded chunks, so the walks a tier hands on are exactly those it handed on before
    // (`n0 != 0` is tested on an opaque scalar copy in each path: shared between the two, the compiler carries the condition through
    // a vector register -- v_cndmask, v_cmp_ne, s_and, s_cbranch_vccnz)
    uint32_t n0a = n0, n0b = n0;
    asm volatile("" : "+s"(n0a));
    asm volatile("" : "+s"(n0b));
    if (ADD && hcount + deg <= ws.hlimit && c + deg <= ws.cap) {
        if (n0a) scan_chunk<64, SP, ADD, STG, false>(ws, g, v, root_vi, size, c, hcount, ecount, sc, e0, start + (uint32_t)g.lane - 1u);
        const uint32_t r1 = start + deg;
        for (uint32_t base = start + inl; base < r1; base += 64) {
            const uint32_t p = base + (uint32_t)g.lane;
            int2 e = UGS_NO_ENTRY;
            if (p < r1) e = P.adj[p];
            scan_chunk<64, SP, ADD, STG, false>(ws, g, v, root_vi, size, c, hcount, ecount, sc, e, p);
        }
        return true;
    }
    if (n0b && !scan_chunk<64, SP, ADD, STG>(ws, g, v, root_vi, size, c, hcount, ecount, sc, e0, start + (uint32_t)g.lane - 1u)) return false;
    const uint32_t r1 = start + deg;
    for (uint32_t base = start + inl; base < r1; base += 64) {
        const uint32_t p = base + (uint32_t)g.lane;
        int2 e = UGS_NO_ENTRY;
        if (p < r1) e = P.adj[p];
        if (!scan_chunk<64, SP, ADD, STG>(ws, g, v, root_vi, size, c, hcount, ecount, sc, e, p)) return false;
    }
    return true;
}

// End of a complete walk: the <= 32 hits become the row's directed items (hit at position p of row s pointing to member i:
// item s->i, and its mirror i->s unless i == s), ranked in the output order -- source index, then CSR position, and inside
// one row CSR position order is edge-column order (the symmetrised CSR is built in column order; equal columns only for
// the two identical entries of a self loop).  One lane per hit, the others' keys come through v_readlane as scalars, four hits
// per trip (the loop control is scalar work per trip; lanes past the hits hold keys that sort behind every real one, so the extra
// compares of the last trip add nothing).  Columns tie only between the two entries of a self loop: the tie-breaking second
// compare of the column ranks runs only when the row has a self hit at all (a wave-uniform test).
// (Round 4 also measured item ranks by COUNTING -- every item sets bit cr of its source's word in LDS, a rank is the population of
// the smaller sources' words plus the lower bits of its own: ~170 instructions fewer per walk, but three more LDS round trips at
// the very end of the walk: 5.06 against 4.94 ms per 1M walks, not kept.)
__device__ __forceinline__ void stage_flush(uint32_t ne, const Grp<64> &g, const uint32_t *SV, uint32_t k, uint4 en, uint32_t ecol, uint2 *out) {
    const uint32_t lane = (uint32_t)g.lane;
    const bool mine = lane < ne;
    uint32_t ei = 0u;
    for (uint32_t j = 0; j < k; ++j) ei = (SV[j] == en.y) ? j : ei;
    const uint32_t es = en.z;
    uint32_t cr = 0u;
    const uint32_t ecol_s = mine ? ecol : 0xFFFFFFFFu;
    if (g.any(mine && ei == es)) {
        for (uint32_t t = 0; t < ne; t += 4) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t ct = g.bcast(ecol_s, (int)((t + u) & 63u));
                cr += (ct < ecol || (ct == ecol && t + u < lane)) ? 1u : 0u;
            }
        }
    } else {
        for (uint32_t t = 0; t < ne; t += 4) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) cr += g.bcast(ecol_s, (int)((t + u) & 63u)) < ecol ? 1u : 0u;
        }
    }
    const bool mirror = mine && ei != es;
    const uint32_t ka = es * UGS_STAGE_ENTRIES + cr, kb = ei * UGS_STAGE_ENTRIES + cr;
    const uint32_t kk = (ka << 16) | kb;
    uint32_t ra = 0u, rb = 0u;
    const uint32_t kk_s = mine ? kk : 0xFFFFFFFFu;
    for (uint32_t t = 0; t < ne; t += 4) {
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint32_t kt = g.bcast(kk_s, (int)((t + u) & 63u));
            const uint32_t kat = kt >> 16, kbt = ((kt >> 21) == ((kt >> 5) & 0x7FFu)) ? 0xFFFFu : (kt & 0xFFFFu);
            ra += (kat < ka ? 1u : 0u) + (kbt < ka ? 1u : 0u);
            rb += (kat < kb ? 1u : 0u) + (kbt < kb ? 1u : 0u);
        }
    }
    if (mine) out[ra] = make_uint2(ecol, es | (ei << 8));
    if (mirror) out[rb] = make_uint2(ecol, ei | (es << 8));
}

// One walk.  Returns false on workspace overflow (the row is then redone by the next tier).
// (Round 4 measured a block-wide LDS copy of the walks' graph for the 8-lane tier -- row pointer, adjacency and root records of the
// graph that owns the block's 32 consecutive rows: PROTEINS-shaped walk 27.8 against 27.0 us, QM9-shaped 59 against 45 us (two blocks
// per CU instead of three).  An 8-lane walk is a chain of LDS round trips of its own order stages; the two L2 round trips per step it
// saved do not show.  The same copy for the fill kernels (row pointer and (neighbour, column) entries): 12.4 against 12.4 us and 25.5
// against 25.0 us.  Neither kept.)
template <int GS, class SP, int MAXPER, bool PAD>
__device__ __forceinline__ bool do_walk(const Work<SP> &ws, const Grp<GS> &g_, const UgsWalkArgs &a, int64_t row_rel,
                                        uint32_t *SV /* [UGS_KMAX] group-private */, uint4 *EL /* [UGS_STAGE_ENTRIES] or null */,
                                        uint32_t *nedges_out = nullptr /* the row's edge-entry count (0 if the walk is handed on) */) {

    static_assert(!PAD || GS == 64, "padded rows are read by a whole wave");
    const Grp<GS> &g = g_;
    constexpr bool STG = GS == 64 && sizeof(typename SP::TW) == 4;             // one walk per wave, LDS workspace
    StageCtx sc;
    sc.EL = EL; sc.ne = 0u; sc.on = STG && a.stage != nullptr && EL != nullptr; sc.onm = 0ull;
    if constexpr (STG) {      // (opaque, or the compiler turns `hits & onm` back into two tests)
        uint32_t on32 = g.uni(sc.on ? 1u : 0u);
        asm volatile("" : "+s"(on32));
        sc.onm = 0ull - (uint64_t)on32;
    }
    const UgsPlanDev &P = a.plan;
    const int64_t row = a.row_begin + row_rel;
    int64_t gi, i;
    if (P.num_graphs == 1) { gi = 0; i = row; }
    else { gi = row / a.m; i = row - gi * a.m; }
    const UgsGraphDesc gd = P.graphs[gi];
    const int k = a.k;
    int64_t *out = a.nodes + row_rel * k;
    if (nedges_out) *nedges_out = 0u;
    if (gd.level < 0) {   // degenerate graph: m rows of -1, no edges (reference src/ugs_sampler_batch_extension.cpp:132-143)
        for (int j = g.lane; j < k; j += GS) out[j] = -1;
        if (g.lane == 0) a.counts[row_rel] = 0;
        return true;
    }
    STAMP_DECL;
    STAMP_BEGIN();
    Rng rng;
    rng.init((a.seed_ptr ? *a.seed_ptr : a.seed64) + (uint64_t)i * 0x9e3779b97f4a7c15ull);
    // Root draw.  The root record is fetched first and the membership hash is reset while it is in flight; the root's row is
    // requested as soon as the root is known, before the root is entered into the hash and the sample list.
    uint32_t root_vi, root_v;
    uint32_t j = 0;
    double u = 0.0;
    UgsRootRec rr{};
    int2 vr = make_int2(0, 0);
    if (gd.level == 0) {      // alias draw: two numbers (reference include/sampler.hpp:72-77)
        j = mod64_root(rng.next(), (uint32_t)gd.n);
        u = (double)rng.next() * 0x1p-64;           // == / (double)UINT64_MAX (which is 2^64): exact scaling
        rr = P.roots[gd.vbase + j];
    } else {                  // relaxed: uniform over the viable list, one number (reference src/sampler.cpp:169-172)
        const uint32_t idx = mod64_root(rng.next(), (uint32_t)gd.n_viable);
        vr = P.viable[gd.viable_base + idx];
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // reset the membership hash (16 bytes per store)
        uint4 *H4 = reinterpret_cast<uint4 *>(ws.HK);
        const uint32_t n4 = (ws.hmask + 1u) >> 2;
        for (uint32_t s = g.lane; s < n4; s += GS) H4[s] = make_uint4(kEmpty, kEmpty, kEmpty, kEmpty);
    }
    if (gd.level == 0) {
        const bool self = u < rr.prob;
        root_vi = g.uni(self ? j : (uint32_t)rr.alias);
        root_v = g.uni((uint32_t)(self ? rr.v_self : rr.v_alias));
    } else {
        root_vi = g.uni((uint32_t)vr.x);
        root_v = g.uni((uint32_t)vr.y);
    }
    uint32_t r0 = 0, r1 = 0;
    int2 e0 = make_int2(0, 0);
    uint32_t v = root_v;                                                      // the vertex whose row is scanned next (local index size-1)
    if constexpr (PAD) e0 = load_prow(P, gd.vbase + v, g.lane);
    else { r0 = g.uni((uint32_t)P.rowptr[gd.rbase + v]); r1 = g.uni((uint32_t)P.rowptr[gd.rbase + v + 1]); }
    SP::sync();
    if (g.lane == 0) { ws.HK[hash_slot(root_v, ws.hmask)] = root_v | kInS; SV[0] = root_v; }
    SP::sync();
    uint32_t size = 1, c = 0, hcount = 1, ecount = 0;
    int nvalid = 0;           // leading stages of the order computation that are still valid
    STAMP_END(0);
    for (int step = 0;; ++step) {
        Grp<GS> g = g_;                                                       // see select_lds: keeps lane-derived constants out of long-lived registers
        if constexpr (GS == 64) asm volatile("" : "+v"(g.lane));
#ifdef UGS_PAD      // diagnostic A/B builds only (tools/pad_probe.sh): 64 dummy instructions per growth step -- scalar or vector, 4- or 8-byte
        {           // encodings -- to tell instruction-issue, instruction-fetch and vector-pipe limits apart; no output depends on them
            uint32_t pad_;
#if UGS_PAD == 1
#define UGS_PAD1 asm volatile("s_mov_b32 %0, 0" : "=s"(pad_));
#elif UGS_PAD == 2
#define UGS_PAD1 asm volatile("s_mov_b32 %0, 0x12345678" : "=s"(pad_));
#elif UGS_PAD == 3
#define UGS_PAD1 asm volatile("v_mov_b32_e32 %0, 0" : "=v"(pad_));
#elif UGS_PAD == 4
#define UGS_PAD1 asm volatile("v_mov_b32_e64 %0, 0" : "=v"(pad_));
#elif UGS_PAD == 5      /* a conditional branch that is not taken */
#define UGS_PAD1 asm volatile("s_cmp_eq_u32 0, 0\ns_cbranch_scc0 ugs_pad_%=\nugs_pad_%=:" : "=s"(pad_) : : "scc");
#elif UGS_PAD == 6      /* a taken branch (to the next instruction) */
#define UGS_PAD1 asm volatile("s_branch ugs_pad_%=\nugs_pad_%=:" : "=s"(pad_));
#elif UGS_PAD == 7      /* the wait-state filler the compiler puts in front of DPP reads */
#define UGS_PAD1 asm volatile("s_nop 0" : "=s"(pad_));
#elif UGS_PAD == 8
#define UGS_PAD1 asm volatile("s_nop 1" : "=s"(pad_));
#else                   /* a counter wait with nothing outstanding */
#define UGS_PAD1 asm volatile("s_waitcnt lgkmcnt(0)" : "=s"(pad_));
#endif
#define UGS_PAD8 UGS_PAD1 UGS_PAD1 UGS_PAD1 UGS_PAD1 UGS_PAD1 UGS_PAD1 UGS_PAD1 UGS_PAD1
            UGS_PAD8 UGS_PAD8 UGS_PAD8 UGS_PAD8 UGS_PAD8 UGS_PAD8 UGS_PAD8 UGS_PAD8
#undef UGS_PAD8
#undef UGS_PAD1
        }
#endif
        bool ok;
        if (step < k - 1) {                                                   // the last vertex adds no candidates
            if constexpr (PAD) ok = scan_prow<SP, true, STG>(ws, g, P, v, root_vi, size, c, hcount, ecount, e0, gd.vbase + v, sc);
            else ok = scan_row<GS, SP, true, STG>(ws, g, P.adj, v, root_vi, size, c, hcount, ecount, r0, r1, sc);
        } else {
            if constexpr (PAD) ok = scan_prow<SP, false, STG>(ws, g, P, v, root_vi, size, c, hcount, ecount, e0, gd.vbase + v, sc);
            else ok = scan_row<GS, SP, false, STG>(ws, g, P.adj, v, root_vi, size, c, hcount, ecount, r0, r1, sc);
        }
        STAMP_END(1);
        if (!ok) return false;
        if (step >= k - 1 || c == 0) break;                                   // complete, or growth failed: partial row
        const uint32_t rsel = g.uni(mod64_by<sizeof(typename SP::TW) == 4, sizeof(typename SP::TW) == 4 && GS == 64>(rng.next(), c));
        STAMP_END(4);
        const Pick pick = select_any<GS, MAXPER>(ws, g, c, rsel, nvalid);
        const uint32_t w = pick.w;
        if constexpr (PAD) {
            e0 = load_prow(P, gd.vbase + w, g.lane);                  // the row itself: issued now, consumed after the candidate list has been updated
        } else {
            r0 = (uint32_t)P.rowptr[gd.rbase + w];
            r1 = (uint32_t)P.rowptr[gd.rbase + w + 1];
            r0 = g.uni(r0); r1 = g.uni(r1);
        }
        STAMP_END(2);
        // move w from the candidates to the sample: drop it from D keeping the order of the others.  The LDS tiers' final stage
        // names its position; the global-memory tier searches D.
        uint32_t q = pick.q;
        if constexpr (sizeof(typename SP::TW) != 4) {
            for (uint32_t t0 = 0; t0 < c && q == c; t0 += 4 * GS) {    // 4 chunks per round trip
                uint32_t x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t t = t0 + u * GS + g.lane; x[u] = (t < c) ? ws.D[t] : kEmpty; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint64_t mk = g.ballot(x[u] == w);
                    if (mk && q == c) q = t0 + u * GS + (uint32_t)(__ffsll((long long)mk) - 1);
                }
            }
        }
        STAMP_END(7);
        for (uint32_t t0 = q; t0 + 1 < c; t0 += 4 * GS) {
            uint32_t x[4];
            if constexpr (sizeof(typename SP::TW) == 4) {
                // LDS tiers: nothing predicated.  Reads past the candidates stay inside the walk's workspace (the order arrays and
                // the bucket table follow D); writes past the new end land in dead slots of D (index clamped to its last one).
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = ws.D[t0 + u * GS + g.lane + 1];
                SP::sync();
                {   // (the clamp against cap-1-u*GS on the un-offset index: u*GS rides in the write's offset field, one v_min per write)
                    const uint32_t tb = t0 + (uint32_t)g.lane;
                    constexpr uint32_t last = (uint32_t)(MAXPER * GS) - 1u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t lim4 = (last - (uint32_t)(u * GS)) * 4u, tb4 = tb * 4u;                      // byte offsets: no shift after the clamp
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ws.D + u * GS) + (tb4 < lim4 ? tb4 : lim4)) = x[u];
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t t = t0 + u * GS + g.lane; x[u] = (t + 1 < c) ? ws.D[t + 1] : 0u; }
                SP::sync();
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t t = t0 + u * GS + g.lane; if (t + 1 < c) ws.D[t] = x[u]; }
            }
            SP::sync();
        }
        {   // stages whose candidates all precede position q keep their order
            int keep = 0;
            if constexpr (sizeof(typename SP::TW) == 4) {                     // LDS tiers: the chain values are immediates
                if constexpr (GS == 64) keep = chain_index_below_lanes(q + 1u, g_.chain); else
                keep = chain_index_below<nst_of(MAXPER * GS)>(q + 1u);        // number of B[i] <= q
                keep = keep < nvalid ? keep : nvalid;
            } else {
                while (keep < nvalid && d_chain.B[keep] <= q) ++keep;
            }
            nvalid = keep;
        }
        c -= 1;
        STAMP_END(8);
        {   // w is now a member of the sample: flag its hash entry (the group probes GS consecutive slots per round trip)
            const uint32_t step = hash_step(w);
            // lane i looks at the i-th slot of w's probe sequence; the entry is there (w was a candidate) and almost always among
            // the first GS slots
            if constexpr (GS == 64 && sizeof(typename SP::TW) == 4) {
                // one walk per wave: the first round against EXEC -- read, v_cmpx (the lane that sees the entry stays), write under that
                // mask (nobody writes when nobody saw it), and the mask itself says whether a second round is needed: seven
                // instructions and no taken branch where the compiler's loop took sixteen and two
                uint32_t ls;                                           // lane * step (< 2^18): the full-rate 24-bit multiply, by hand -- the
                asm("v_mul_u32_u24 %0, %1, %2" : "=v"(ls) : "s"(step), "v"(g.lane));   // lane index is opaque to the compiler (do_walk), which then takes v_mul_lo_u32
                uint32_t s = (hash_slot(w, ws.hmask) + ls) & ws.hmask;
                const uint32_t addr = (uint32_t)(uintptr_t)ws.HK + (s << 2);
                uint32_t t;
                uint64_t saved, found;
                asm volatile(
                    "s_mov_b64 %[sv], exec\n"
                    "ds_read_b32 %[t], %[addr]\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "v_cmpx_eq_u32 %[w], %[t]\n"
                    "ds_write_b32 %[addr], %[wf]\n"
                    "s_mov_b64 %[fd], exec\n"
                    "s_mov_b64 exec, %[sv]\n"
                    : [t] "=&v"(t), [sv] "=&s"(saved), [fd] "=&s"(found)
                    : [addr] "v"(addr), [w] "s"(w), [wf] "v"(w | kInS)
                    : "vcc", "memory");
                if (__builtin_expect(found == 0ull, 0)) {             // deeper than 64 probes: the general loop
                    bool hit = false;
                    for (uint32_t it = GS; !g.any(hit) && it <= ws.hmask; it += GS) {
                        s = (s + GS * step) & ws.hmask;
                        hit = ws.HK[s] == w;
                    }
                    if (hit) ws.HK[s] = w | kInS;
                }
            } else {
            uint32_t s = (hash_slot(w, ws.hmask) + (uint32_t)g.lane * step) & ws.hmask;
            bool hit = ws.HK[s] == w;                                 // a candidate's entry carries no flag
            for (uint32_t it = GS; __builtin_expect(!g.any(hit), 0) && it <= ws.hmask; it += GS) {
                s = (s + GS * step) & ws.hmask;
                hit = ws.HK[s] == w;
            }
            if (hit) ws.HK[s] = w | kInS;
            }
            if constexpr (GS == 64) SV[size] = w;                     // every lane, same word, same value: no lane-0 mask to set up
            else
            if (g.lane == 0) SV[size] = w;
        }
        size += 1;
        v = w;
        SP::sync();
        STAMP_END(3);
    }
    if constexpr (GS == 64) ecount = g.last(g.prefix_incl(ecount));              // the lanes' counts -> the row's
    const uint32_t nedges = (size == (uint32_t)k) ? ecount : 0u;                // incomplete rows carry no edges (:219-223)
    // staged hits: fetch their edge columns now, the row's epilogue below runs while the gather is in flight
    bool flush = false;
    uint4 en = make_uint4(0u, 0u, 0u, 0u);
    uint32_t ecol = 0u;
    if constexpr (STG) {
        flush = sc.on && nedges != 0u && sc.ne <= UGS_STAGE_ENTRIES;
                if (flush && (uint32_t)g.lane < sc.ne) { en = sc.EL[g.lane]; ecol = (uint32_t)P.adjf[en.x].y; }
    }
    // nodes row: growth order, -1 padded (reference src/sampler.cpp:205-216, src/ugs_sampler_batch_extension.cpp:188-196)
    const int64_t off = gd.node_lo + a.extra_node_off;
    for (int j = g.lane; j < k; j += GS) out[j] = (j < (int)size) ? (int64_t)SV[j] + off : (int64_t)-1;
    // one word per row: the edge-entry count and, in its top bit, whether the row's items are staged (one store instead of two)
    if (g.lane == 0) a.counts[row_rel] = nedges | (flush ? UGS_COUNT_STAGED : 0u);
    if (nedges_out) *nedges_out = nedges;
    if (a.stage) {                                                               // staging is on for this call
        if constexpr (STG) { if (flush) stage_flush(sc.ne, g, SV, (uint32_t)k, en, ecol, a.stage + row_rel * UGS_STAGE_ITEMS); }
        if (g.lane == 0 && !flush && nedges != 0u) a.ulist[atomicAdd(a.ucount, 1u)] = row_rel;
    }
    STAMP_END(5);
    return true;
}

// LDS words of one group's workspace for a tier (all sub-arrays 16-byte aligned)
template <int CAP> struct TierCfg {
    static constexpr int NSTAGE = nst_of(CAP);                                     // stages that are ever materialised
    static constexpr int ORDW = (int)(((ord_words_before(NSTAGE) + 1u) / 2u + 3u) & ~3u);   // 16-bit positions: 52 / 248 / 520 / 1072 words
    static constexpr int TI = tier_index(CAP);
    static constexpr int BCAP_A = tbl_words(CAP);                                   // bucket-table words (see tbl_words)
    static constexpr int HS = CAP <= 32 ? 64 : (TI == 0 ? 128 : (TI == 1 ? 512 : (TI == 2 ? 1024 : (TI <= 4 ? 2048 : 4096))));
    static constexpr int HLIMIT = (TI == 1 || TI == 2 || TI == 4) ? HS / 8 * 7 : HS / 4 * 3;   // max distinct vertices a walk may have seen
    static_assert(CAP <= 64 || BCAP_A * 4 >= 127 * 16, "mates2_by_table keeps 16 bytes per bucket of the 127-bucket stage in TBL");
    // (CAP 32 is a form of tier S the host picks only for walks that cannot outgrow it: UGS_SMALL_CAP / UGS_SMALL_HASH_LIMIT)
    static_assert(CAP == UGS_SMALL_CAP ? HLIMIT == UGS_SMALL_HASH_LIMIT : (HLIMIT == UGS_TIER_HASH_LIMIT[TI] && CAP == UGS_TIER_CAP[TI]),
                  "host tier logic (choose_tier) relies on these limits");
    static constexpr int ELW = CAP > 64 ? 4 * UGS_STAGE_ENTRIES : 0;             // staged hits (one-walk-per-wave tiers)
    static constexpr int WORDS = CAP /*D*/ + ORDW + BCAP_A /*TBL*/ + HS /*HK*/ + UGS_KMAX /*SV*/ + ELW;
};

// second launch-bounds argument = waves per SIMD the register allocation must allow.  CAP 448: 5 (96 VGPRs, 3 spilled).
// Measured on C5 (same-box A/B, census build for residency): the LDS is granted in 1280-byte granules, so 8.6 KB/walk
// admits 18 one-wave blocks per CU (5,5,4,4 per SIMD).  With a STATIC split of the rows 18 blocks/CU was slower than 16
// (10.63 vs 10.43 ms: a launch ended with the waves of the fuller SIMDs); with the shared work counter the extra waves are
// pure throughput: 8.81 -> 8.51 ms.  Spilling further to reach more waves costs more than it brings (30 % in an early build).
template <int GS, int CAP, int BLOCK, bool PAD>
__global__ __launch_bounds__(BLOCK, CAP <= 32 ? 4 : ((CAP > 64 && CAP <= 512) ? 5 : (CAP == 704 ? 3 : (CAP <= 64 || CAP == 1024 || CAP == 1408 ? 2 : 1)))) void ugs_walk_lds(UgsWalkArgs a) {
    using Cfg = TierCfg<CAP>;
    constexpr int GROUPS = BLOCK / GS;
    __shared__ __attribute__((aligned(16))) uint32_t lds[GROUPS * Cfg::WORDS];
    Grp<GS> g;
    g.init();
    if constexpr (GS == 64) g.chain = chain_lane_const<Cfg::NSTAGE>(g.lane);
    const int gib = (int)threadIdx.x / GS;
    uint32_t *base = lds + gib * Cfg::WORDS;
    Work<LdsSpace> ws;
    ws.D = base;
    ws.ORD = reinterpret_cast<uint16_t *>(ws.D + CAP);
    ws.AUX = nullptr;                      // the register-resident stages need no per-element scratch
    ws.TBL = ws.D + CAP + Cfg::ORDW;
    ws.HK = ws.TBL + Cfg::BCAP_A;
    uint32_t *SV = ws.HK + Cfg::HS;
    uint4 *EL = Cfg::ELW ? reinterpret_cast<uint4 *>(SV + UGS_KMAX) : nullptr;
    ws.cap = CAP;
    ws.hmask = Cfg::HS - 1;
    ws.hlimit = Cfg::HLIMIT;
#ifdef UGS_STAMPS
    __shared__ unsigned long long stamps[32];
    if (threadIdx.x < 32) stamps[threadIdx.x] = 0ull;
    __syncthreads();
    ws.ST = stamps;
    struct Flush { unsigned long long *st; __device__ ~Flush() { __syncthreads(); if (threadIdx.x < 32) atomicAdd(&ugs_stamp_buffer[threadIdx.x], st[threadIdx.x]); } } flush_{stamps};
#endif
    const int64_t total = a.in_list ? (int64_t)*a.in_count : a.row_count;
    const int64_t ngroups = (int64_t)gridDim.x * GROUPS;
    // Work distribution.  Dynamic (a.work_next, launches with many more walks than resident groups): chunks of UGS_WORK_CHUNK
    // consecutive items, the first chunk by group index, every further one from a device counter (one round trip per chunk: a
    // single hot address answers in several microseconds); a walk's cost varies, so a static split ends with the unluckiest wave.
    // Two loops, two inlined copies of the walk: one shared loop measured 0.8 % slower on C5 (register allocation).
    if (a.work_next) {
        // The counter counts ITEMS: a group takes its first chunk by index, every further one by atomicAdd(counter, chunk) with a
        // chunk that shrinks as the launch drains -- 4 while more than 8 items per group are left (2 / 4 / 8 / 16 rows per counter
        // round trip measured 8.85 / 8.83 / 8.91 / 9.04 ms per 1M walks; 1 throughout is 50 % slower: the counter's round trip
        // under contention), then 2, and 1 for the last two items per group, so that the launch ends within one walk of even
        // (guided self-scheduling; matters most for launches that give a group only a few dozen walks, e.g. a rank's shard).
        // (against fixed chunks of 4, or 2 for small launches: 6.80 -> 6.77 ms per 1M walks, 0.951 -> 0.929 ms per 125k)
        auto chunk_for = [&](int64_t left) -> int64_t { return left > 8 * ngroups ? 4 : (left > 2 * ngroups ? 2 : 1); };
        const int64_t first = chunk_for(total);
        int64_t it = ((int64_t)blockIdx.x * GROUPS + gib) * first, end = it + first;
        while (it < total) {
            if (end > total) end = total;
            for (; it < end; ++it) {
                const int64_t row_rel = a.in_list ? a.in_list[it] : it;
                if (!do_walk<GS, LdsSpace, (CAP + GS - 1) / GS, PAD>(ws, g, a, row_rel, SV, EL)) {
                    if (g.lane == 0 && a.ovf_list) { uint32_t pos = atomicAdd(a.ovf_count, 1u); a.ovf_list[pos] = row_rel; }
                }
            }
            const int64_t chunk = chunk_for(total - it);                 // `it` is close to the counter: the estimate only picks the chunk size
            unsigned long long nxt = 0ull;
            if (g.lane == 0) nxt = atomicAdd(a.work_next, (unsigned long long)chunk);
            const uint32_t lo = g.bcast((uint32_t)nxt, 0), hi = g.bcast((uint32_t)(nxt >> 32), 0);
            it = ngroups * first + (int64_t)(((unsigned long long)hi << 32) | lo);
            end = it + chunk;
        }
        return;
    }
    if constexpr (GS == 8 || GS == 16) {
        if (a.wsum) {
            // Small-batch step (ugs_plan_step): beside the per-row counts the walk leaves the SUM of every 8 consecutive rows -- the
            // groups of a wave hold consecutive rows and come back from their walks together -- so that the fill kernel can add
            // up what lies in front of a tile from plain, cacheable words written by the kernel BEFORE it (no communication between
            // the fill's blocks: that cost 15 us on the QM9-shaped batch, see ugs_fill_scan).  Rows are taken by index here (no
            // list of handed-on rows: the host asks for the sums only when no walk can be handed on).
            constexpr int RPW = 64 / GS;                                     // rows per wave: 8, or 4 (two waves make a sum)
            __shared__ uint32_t wave_sum[BLOCK / 64];
            for (int64_t it0 = (int64_t)blockIdx.x * GROUPS; it0 < total; it0 += ngroups) {
                const int64_t it = it0 + gib;
                uint32_t ne = 0u;
                if (it < total) (void)do_walk<GS, LdsSpace, (CAP + GS - 1) / GS, PAD>(ws, g, a, it, SV, EL, &ne);
                uint32_t sum = g.lane == 0 ? ne : 0u;
#pragma unroll
                for (int d = GS; d < 64; d <<= 1) sum += __shfl_xor(sum, d, 64);
                if constexpr (RPW == 8) {
                    const int64_t w0 = it0 + (int64_t)((threadIdx.x >> 6) * 8);      // first row of this wave
                    if ((threadIdx.x & 63) == 0 && w0 < total) a.wsum[w0 >> 3] = sum;
                } else {
                    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = sum;
                    __syncthreads();
                    if (threadIdx.x < GROUPS / 8) {
                        const int64_t r0 = it0 + (int64_t)threadIdx.x * 8;           // (it0 is a multiple of the block's 16 rows)
                        if (r0 < total) a.wsum[r0 >> 3] = wave_sum[2 * threadIdx.x] + wave_sum[2 * threadIdx.x + 1];
                    }
                    __syncthreads();
                }
            }
            return;
        }
    }
    for (int64_t it = (int64_t)blockIdx.x * GROUPS + gib; it < total; it += ngroups) {
        const int64_t row_rel = a.in_list ? a.in_list[it] : it;
        if (!do_walk<GS, LdsSpace, (CAP + GS - 1) / GS, PAD>(ws, g, a, row_rel, SV, EL)) {
            if (g.lane == 0 && a.ovf_list) { uint32_t pos = atomicAdd(a.ovf_count, 1u); a.ovf_list[pos] = row_rel; }
        }
    }
}

// Padded rows (ugs_device.h): entry l of the block of plan vertex R.  One thread per entry; runs once per plan.
__global__ __launch_bounds__(256) void ugs_build_prow(UgsPlanDev P, int64_t num_vertices, int2 *prow, int shift) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t R = idx >> shift;
    if (R >= num_vertices) return;
    const int l = (int)(idx & ((1 << shift) - 1));
    int64_t lo = 0, hi = P.num_graphs - 1;                      // last graph whose first vertex is <= R (degenerate graphs own none)
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (P.graphs[mid].vbase <= R) lo = mid; else hi = mid - 1;
    }
    const int64_t rb = P.graphs[lo].rbase, v = R - P.graphs[lo].vbase;
    const int64_t start = P.rowptr[rb + v], deg = P.rowptr[rb + v + 1] - start;
    int2 e = make_int2(0, 0);
    if (l == 0) e = make_int2((int)deg, (int)start);
    else if (l - 1 < deg) e = P.adj[start + l - 1];
    prow[idx] = e;
}

// last tier: workspace in global memory, one wave per walk, any candidate-set size up to gcap
__global__ __launch_bounds__(64) void ugs_walk_global(UgsWalkArgs a) {
    __shared__ uint32_t SV[UGS_KMAX];
    Grp<64> g;
    g.init();
    uint32_t *base = a.gws + (int64_t)blockIdx.x * a.gws_words_per_group;
    Work<GlbSpace> ws;
    auto al4 = [](int64_t x) { return (x + 3) & ~3ll; };
    ws.TBL = (unsigned long long *)base;                         // 8-byte words first
    ws.D = base + 2 * al4(a.gbcap);
    ws.ORD = ws.D + al4(a.gcap);
    ws.AUX = ws.ORD + al4(a.gpcap);                              // gpcap = words of all materialised stage orders
    ws.HK = ws.AUX + al4(a.gcap);
    ws.cap = (uint32_t)a.gcap;
    ws.hmask = (uint32_t)a.ghs - 1u;
    ws.hlimit = (uint32_t)a.ghs / 4u * 3u;
#ifdef UGS_STAMPS
    __shared__ unsigned long long stamps[32];      // the global tier's phases are not reported
    ws.ST = stamps;
#endif
    const int64_t total = a.in_list ? (int64_t)*a.in_count : a.row_count;
    for (int64_t it = blockIdx.x; it < total; it += gridDim.x) {
        const int64_t row_rel = a.in_list ? a.in_list[it] : it;
        if (!do_walk<64, GlbSpace, 0, false>(ws, g, a, row_rel, SV, nullptr)) {
            // cannot happen when gcap covers the graph's bound; mark the row so the host can report it
            if (g.lane == 0 && a.ovf_list) { uint32_t pos = atomicAdd(a.ovf_count, 1u); a.ovf_list[pos] = row_rel; }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// exclusive scan of the per-row counts -> edge_ptr[rows + 1]
// ------------------------------------------------------------------------------------------------------------------
constexpr int kScanBlock = 256, kScanPer = 8, kScanTile = kScanBlock * kScanPer;

__device__ __forceinline__ int64_t block_excl_scan(int64_t x, int64_t *total, int64_t *sh /* [kScanBlock/64] */) {
    // inclusive scan inside the wave, then across the 4 waves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { long long y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    int64_t woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kScanBlock / 64; ++i) { if (i < wv) woff += sh[i]; tot += sh[i]; }
    __syncthreads();
    *total = tot;
    return woff + incl - x;
}

__global__ __launch_bounds__(kScanBlock) void ugs_scan_partials(const uint32_t *counts, int64_t rows, int64_t *block_sums) {
    __shared__ int64_t sh[kScanBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) if (base + j < rows) s += counts[base + j] & ~UGS_COUNT_STAGED;
    int64_t tot;
    block_excl_scan(s, &tot, sh);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanBlock) void ugs_scan_block_sums(int64_t *block_sums, int64_t nblocks) {
    __shared__ int64_t sh[kScanBlock / 64];
    int64_t carry = 0;
    for (int64_t base = 0; base < nblocks; base += kScanBlock) {
        const int64_t idx = base + threadIdx.x;
        int64_t x = idx < nblocks ? block_sums[idx] : 0, tot;
        int64_t ex = block_excl_scan(x, &tot, sh);
        if (idx < nblocks) block_sums[idx] = carry + ex;
        carry += tot;
    }
}

// the thread that writes the total may also hand it to the host: total, fence, then the launch's epoch into a word of pinned host memory
// (the host polls the word: ugs_host.cpp wait_signal)
struct ScanSignal { int64_t *h_total; uint32_t *h_flag; uint32_t epoch; };
__device__ __forceinline__ void scan_signal(const ScanSignal &sg, int64_t total) {
    if (sg.h_total) { *sg.h_total = total; __threadfence_system(); *(volatile uint32_t *)sg.h_flag = sg.epoch; }
}

__global__ __launch_bounds__(kScanBlock) void ugs_scan_final(const uint32_t *counts, int64_t rows, const int64_t *block_offs,
                                                             int64_t *edge_ptr, ScanSignal sg) {
    __shared__ int64_t sh[kScanBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    uint32_t v[kScanPer];
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) { v[j] = (base + j < rows) ? (counts[base + j] & ~UGS_COUNT_STAGED) : 0u; s += v[j]; }
    int64_t tot;
    int64_t ex = block_excl_scan(s, &tot, sh) + (block_offs ? block_offs[blockIdx.x] : 0);
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) { if (base + j < rows) edge_ptr[base + j] = ex; ex += v[j]; }
    if (rows - 1 >= base && rows - 1 < base + kScanPer) { edge_ptr[rows] = ex; scan_signal(sg, ex); }   // owner of the last row writes the total
}

// Two launches instead of three for up to 4096 tiles (8M rows): every block of the final pass sums the tile totals in front of
// its own tile itself (at most 16 loads per thread and one block reduction; no flags, no waiting)
__global__ __launch_bounds__(kScanBlock) void ugs_scan_final_sum(const uint32_t *counts, int64_t rows, const int64_t *tile_sums,
                                                                 int64_t *edge_ptr, ScanSignal sg) {
    __shared__ int64_t sh[kScanBlock / 64];
    int64_t before = 0, front;
    for (int64_t i = threadIdx.x; i < (int64_t)blockIdx.x; i += kScanBlock) before += tile_sums[i];
    block_excl_scan(before, &front, sh);
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    uint32_t v[kScanPer];
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) { v[j] = (base + j < rows) ? (counts[base + j] & ~UGS_COUNT_STAGED) : 0u; s += v[j]; }
    int64_t tot;
    int64_t ex = block_excl_scan(s, &tot, sh) + front;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) { if (base + j < rows) edge_ptr[base + j] = ex; ex += v[j]; }
    if (rows - 1 >= base && rows - 1 < base + kScanPer) { edge_ptr[rows] = ex; scan_signal(sg, ex); }   // owner of the last row writes the total
}

// single-block variant for small row counts (one launch instead of three): 1024 threads, 8192 rows per round
constexpr int kScanWide = 1024;
__global__ __launch_bounds__(kScanWide) void ugs_scan_small(const uint32_t *counts, int64_t rows, int64_t *edge_ptr, ScanSignal sg) {
    __shared__ int64_t sh[kScanWide / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t carry = 0;
    for (int64_t tile = 0; tile < rows; tile += (int64_t)kScanWide * kScanPer) {
        const int64_t base = tile + (int64_t)threadIdx.x * kScanPer;
        uint32_t v[kScanPer];
        int64_t s = 0;
#pragma unroll
        for (int j = 0; j < kScanPer; ++j) { v[j] = (base + j < rows) ? (counts[base + j] & ~UGS_COUNT_STAGED) : 0u; s += v[j]; }
        long long incl = s;                                          // inclusive scan inside the wave, then across the 16 waves
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { long long y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
        if (lane == 63) sh[wv] = incl;
        __syncthreads();
        int64_t woff = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < kScanWide / 64; ++i) { if (i < wv) woff += sh[i]; tot += sh[i]; }
        __syncthreads();
        int64_t ex = carry + woff + incl - s;
#pragma unroll
        for (int j = 0; j < kScanPer; ++j) { if (base + j < rows) edge_ptr[base + j] = ex; ex += v[j]; }
        carry += tot;
    }
    if (threadIdx.x == 0) { edge_ptr[rows] = carry; scan_signal(sg, carry); }
}

// ------------------------------------------------------------------------------------------------------------------
// fill kernel: induced edges of complete rows, vertex order j then CSR position p (reference src/sampler.cpp:232-243)
// ------------------------------------------------------------------------------------------------------------------
// The k adjacency rows of a sample are FLATTENED into one index space e = 0 .. sum(deg)-1 (row-major = the required
// output order), so every lane's loads are independent: two dependent memory round trips per sample (row bounds, then all
// adjacency entries) instead of two per sampled vertex.
// One complete row: its k adjacency rows flattened, membership tested, the hits compacted in (vertex j, CSR position p) order and
// written with the endpoint numbering of the mode.  The sample's vertices come from the nodes row (`nrow`, batch ids) or, when
// the caller still has them in LDS, from `SVsrc` (graph-local ids).  SV / PS / R0: group-private LDS scratch.
template <int GS>
__device__ __forceinline__ void fill_row(const UgsFillArgs &a, const int64_t *rowptr, const int2 *adjf, const Grp<GS> &g, const UgsGraphDesc &gd, int64_t row_rel, int64_t i,
                                         int64_t e0, const int64_t *nrow, const uint32_t *SVsrc, uint32_t *SV, uint32_t *PS, int64_t *R0) {
    const int k = a.k;
    const int64_t off = gd.node_lo + a.extra_node_off;
    LdsSpace::sync();
    for (int j = g.lane; j < k; j += GS) {
        const uint32_t u = nrow ? (uint32_t)(nrow[j] - off) : SVsrc[j];
        const int64_t r0 = rowptr[gd.rbase + u], r1 = rowptr[gd.rbase + u + 1];
        SV[j] = u;
        R0[j] = r0;
        PS[j + 1] = (uint32_t)(r1 - r0);
    }
    LdsSpace::sync();
    if (g.lane == 0) { uint32_t acc = 0; PS[0] = 0; for (int j = 1; j <= k; ++j) { acc += PS[j]; PS[j] = acc; } }
    LdsSpace::sync();
    const uint32_t T = PS[k];
    int64_t w_off = e0;
    if (k <= 8) {
        // the per-row prefix and vertex lists fit in registers: membership and row lookup are compares on registers
        uint32_t ps[9], sv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { sv[t] = (t < k) ? SV[t] : 0xFFFFFFFEu /* matches no vertex and no idle lane */; ps[t] = (t < k) ? PS[t] : 0xFFFFFFFFu; }
        ps[8] = 0xFFFFFFFFu;
        for (uint32_t cb = 0; cb < T; cb += 4 * GS) {
            uint32_t wv[4];
            int jj[4], ec[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t e = cb + u * GS + g.lane;
                wv[u] = kEmpty; jj[u] = 0; ec[u] = 0;
                if (e < T) {
                    int j = 0;
#pragma unroll
                    for (int t = 1; t < 8; ++t) j += (ps[t] <= e) ? 1 : 0;           // row of flattened entry e
                    uint32_t base_e = ps[0];
#pragma unroll
                    for (int t = 1; t < 8; ++t) base_e = (j == t) ? ps[t] : base_e;
                    jj[u] = j;
                    const int2 nb = adjf[R0[j] + (int64_t)(e - base_e)];          // neighbour and its edge column together
                    wv[u] = (uint32_t)nb.x;
                    ec[u] = nb.y;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (cb + u * GS >= T) break;
                int l = -1;
#pragma unroll
                for (int t = 7; t >= 0; --t) l = (sv[t] == wv[u]) ? t : l;
                const uint64_t mk = g.ballot(l >= 0);
                const int64_t pos = w_off + g.below(mk);
                if (l >= 0 && pos < a.ld) {                           // ld is also the capacity of the caller's edge buffers
                    const int j = jj[u];
                    int64_t uf, vf;
                    if (a.mode == 0) { uf = j; vf = l; }
                    else if (a.mode == 1) { uf = i * k + j; vf = i * k + l; }
                    else if (a.mode == 3) { uf = row_rel * k + j; vf = row_rel * k + l; }     // ids into the flattened nodes of THIS call: what the encoder builds (models/ss_gnn.py:463-464)
                    else { uf = (int64_t)SV[j] + off; vf = (int64_t)SV[l] + off; }
                    a.edge_index[pos] = uf;
                    a.edge_index[a.ld + pos] = vf;
                    a.edge_src[pos] = (int64_t)ec[u];
                }
                w_off += __popcll(mk);
            }
        }
        return;
    }
    for (uint32_t cb = 0; cb < T; cb += 4 * GS) {
        uint32_t wv[4];
        int jj[4], ec[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t e = cb + u * GS + g.lane;
            wv[u] = kEmpty; jj[u] = 0; ec[u] = 0;
            if (e < T) {
                int j = 0;
                for (int t = 1; t < k; ++t) j += (PS[t] <= e) ? 1 : 0;       // row of flattened entry e
                jj[u] = j;
                const int2 nb = adjf[R0[j] + (int64_t)(e - PS[j])];
                wv[u] = (uint32_t)nb.x;
                ec[u] = nb.y;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (cb + u * GS >= T) break;
            int l = -1;
            if (wv[u] != kEmpty) { for (int t = 0; t < k; ++t) if (SV[t] == wv[u]) { l = t; break; } }
            const uint64_t mk = g.ballot(l >= 0);
            const int64_t pos = w_off + g.below(mk);
            if (l >= 0 && pos < a.ld) {
                const int j = jj[u];
                int64_t uf, vf;
                if (a.mode == 0) { uf = j; vf = l; }
                else if (a.mode == 1) { uf = i * k + j; vf = i * k + l; }
                else if (a.mode == 3) { uf = row_rel * k + j; vf = row_rel * k + l; }
                else { uf = (int64_t)SV[j] + off; vf = (int64_t)SV[l] + off; }
                a.edge_index[pos] = uf;
                a.edge_index[a.ld + pos] = vf;
                a.edge_src[pos] = (int64_t)ec[u];
            }
            w_off += __popcll(mk);
        }
    }
}

template <int GS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void ugs_fill(UgsFillArgs a) {
    constexpr int GROUPS = BLOCK / GS;
    __shared__ uint32_t sv_all[GROUPS * UGS_KMAX];
    __shared__ uint32_t ps_all[GROUPS * (UGS_KMAX + 1)];
    __shared__ int64_t r0_all[GROUPS * UGS_KMAX];
    Grp<GS> g;
    g.init();
    const int gib = (int)threadIdx.x / GS;
    uint32_t *SV = sv_all + gib * UGS_KMAX;
    uint32_t *PS = ps_all + gib * (UGS_KMAX + 1);       // PS[j] = entries of rows < j
    int64_t *R0 = r0_all + gib * UGS_KMAX;
    const UgsPlanDev &P = a.plan;
    const int k = a.k;
    const int64_t ngroups = (int64_t)gridDim.x * GROUPS;
    const int64_t todo = a.ulist ? (int64_t)*a.ucount : a.row_count;      // with staging: only the rows the walk could not stage
    for (int64_t it = (int64_t)blockIdx.x * GROUPS + gib; it < todo; it += ngroups) {
        const int64_t row_rel = a.ulist ? a.ulist[it] : it;
        const int64_t e0 = a.edge_ptr[row_rel], e1 = a.edge_ptr[row_rel + 1];
        if (e1 == e0) continue;                               // incomplete or edgeless row
        const int64_t row = a.row_begin + row_rel;
        int64_t gi, i;
        if (P.num_graphs == 1) { gi = 0; i = row; }
        else { gi = row / a.m; i = row - gi * a.m; }
        const UgsGraphDesc gd = P.graphs[gi];
        const int64_t *nrow = a.nodes + row_rel * k;
        fill_row<GS>(a, P.rowptr, P.adjf, g, gd, row_rel, i, e0, nrow, nullptr, SV, PS, R0);
    }
}

// The small-batch step in two launches instead of three: the fill kernel of the 8-lane tier with the exclusive scan of the walk's
// per-row counts folded in (reference semantics of edge_ptr: src/sampler.cpp:249-287).  A block takes tiles of 32 consecutive rows; what
// lies in front of a tile it adds up itself from the sums of 8 rows the WALK kernel left (UgsWalkArgs::wsum: 4 words per tile, at most
// 64 loads per thread for the 131 072 rows this form serves) -- plain cacheable reads of words written by the previous kernel.
// No block talks to another.  Three versions that did were measured first (same outputs, tests/test_gpu_parity.py): tiles handed out by
// tickets and a look-back chain (2048 atomics on one address: 61 us for the fill of the QM9-shaped batch), look-back without tickets
// (every tile of a small batch starts at once, nobody has a prefix yet: 32-40 us), tile sums + sums of groups of 64 tiles (33 us:
// the agent-scope loads and stores that carry the sums between blocks on different XCDs cost 15 us whatever is waited for or not).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void ugs_fill_scan(UgsFillArgs a_in) {
    UgsFillArgs a = a_in;                   // (packed form: ld and edge_src are set once the total is known)
    constexpr int GS = 8, GROUPS = BLOCK / GS;
    static_assert(GROUPS == 32, "a tile is 32 rows = four of the walk's 8-row sums");
    __shared__ uint32_t sv_all[GROUPS * UGS_KMAX];
    __shared__ uint32_t ps_all[GROUPS * (UGS_KMAX + 1)];
    __shared__ int64_t r0_all[GROUPS * UGS_KMAX];
    __shared__ uint32_t cnt_sh[GROUPS];
    __shared__ unsigned long long excl_sh[GROUPS];
    __shared__ unsigned long long part_sh[BLOCK / 64];
    __shared__ unsigned long long all_sh[BLOCK / 64];
    Grp<GS> g;
    g.init();
    const int gib = (int)threadIdx.x / GS;
    const bool packed = a.packed_cap != 0;
    const long long nw = (long long)((a.row_count + 7) / 8);             // the walk's 8-row sums
    bool write = true;
    uint32_t *SV = sv_all + gib * UGS_KMAX;
    uint32_t *PS = ps_all + gib * (UGS_KMAX + 1);
    int64_t *R0 = r0_all + gib * UGS_KMAX;
    const UgsPlanDev &P = a.plan;
    const int k = a.k;
    const long long ntiles = (long long)((a.row_count + GROUPS - 1) / GROUPS);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row_rel = (int64_t)tile * GROUPS + gib;
        const bool in = row_rel < a.row_count;
        const uint32_t c = in ? (a.counts[row_rel] & ~UGS_COUNT_STAGED) : 0u;
        if (g.lane == 0) cnt_sh[gib] = c;
        // everything in front of the tile: the block's threads share the walk's 8-row sums
        // (16-byte loads, eight of them in flight per thread: one word per trip of a rolled loop was one L2 round trip per 256 words --
        // 32 of them in a row for the last tiles of the QM9-shaped batch, most of the kernel's 20 us)
        unsigned long long before = 0ull, all = 0ull;
        const uint4 *w4 = reinterpret_cast<const uint4 *>(a.wsum);
        for (long long t0 = 0; t0 < tile; t0 += 8 * BLOCK) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const long long t = t0 + u * BLOCK + threadIdx.x; v[u] = t < tile ? w4[t] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
            for (int u = 0; u < 8; ++u) before += (unsigned long long)v[u].x + v[u].y + v[u].z + v[u].w;
        }
        if (packed && tile == (long long)blockIdx.x) {                    // packed: the sum of ALL rows too, once per block
            all = before;
            const long long n4 = nw >> 2;
            for (long long t0 = tile; t0 < n4; t0 += 8 * BLOCK) {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const long long t = t0 + u * BLOCK + threadIdx.x; v[u] = t < n4 ? w4[t] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
                for (int u = 0; u < 8; ++u) all += (unsigned long long)v[u].x + v[u].y + v[u].z + v[u].w;
            }
            if ((long long)threadIdx.x < nw - 4 * n4) all += a.wsum[4 * n4 + threadIdx.x];        // the last one to three words
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) all += (unsigned long long)__shfl_xor((long long)all, d, 64);
            if ((threadIdx.x & 63) == 0) all_sh[threadIdx.x >> 6] = all;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) before += (unsigned long long)__shfl_xor((long long)before, d, 64);
        if ((threadIdx.x & 63) == 0) part_sh[threadIdx.x >> 6] = before;
        __syncthreads();
        if (packed && tile == (long long)blockIdx.x) {
            unsigned long long tot = 0ull;
#pragma unroll
            for (int wv = 0; wv < BLOCK / 64; ++wv) tot += all_sh[wv];
            a.ld = (int64_t)tot;                                          // edge_index [2, total], edge_src right behind it
            a.edge_src = a.edge_index + 2 * (int64_t)tot;
            write = 3 * (int64_t)tot <= a.packed_cap;
            // block 0 hands the total to the host right away: the caller allocates its tensors while the rows are being filled
            if (blockIdx.x == 0 && threadIdx.x == 0 && a.h_total) { *a.h_total = (int64_t)tot; __threadfence_system(); *(volatile uint32_t *)a.h_flag = a.epoch; }
        }
        if (threadIdx.x < 64) {                                       // the tile's own 32 counts
            const int lane = (int)threadIdx.x;
            const uint32_t x = lane < GROUPS ? cnt_sh[lane] : 0u;
            uint32_t incl = x;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
            if (lane < GROUPS) excl_sh[lane] = (unsigned long long)(incl - x);
        }
        __syncthreads();
        unsigned long long front = 0ull;
#pragma unroll
        for (int wv = 0; wv < BLOCK / 64; ++wv) front += part_sh[wv];
        const int64_t e0 = (int64_t)(front + excl_sh[gib]);
        if (in && g.lane == 0) {
            a.edge_ptr_out[row_rel] = e0;
            if (row_rel == a.row_count - 1) {
                a.edge_ptr_out[a.row_count] = e0 + (int64_t)c;
                if (a.h_total && !packed) { *a.h_total = e0 + (int64_t)c; __threadfence_system(); *(volatile uint32_t *)a.h_flag = a.epoch; }
            }
        }
        // (QM9-shaped batch, 22.6 us: 9.9 us without the rows' fill -- counts, prefix, edge_ptr and the launch -- and 12.7 us of fill_row's three
        // dependent global reads per row: nodes -> row pointers -> adjacency entries)
        if (in && c != 0u && write) {
            const int64_t row = a.row_begin + row_rel;
            int64_t gi, i;
            if (P.num_graphs == 1) { gi = 0; i = row; }
            else { gi = row / a.m; i = row - gi * a.m; }
            const UgsGraphDesc gd = P.graphs[gi];
            fill_row<GS>(a, P.rowptr, P.adjf, g, gd, row_rel, i, e0, a.nodes + row_rel * k, nullptr, SV, PS, R0);
        }
        __syncthreads();
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------
#ifdef UGS_STAMPS
extern "C" int ugs_debug_read_stamps(unsigned long long *out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(ugs_stamp_buffer), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ugs_stamp_buffer), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

int64_t ugs_scan_tmp_words(int64_t rows) { return (rows + kScanTile - 1) / kScanTile + 1; }

int64_t ugs_global_ws_words(int64_t gcap, int64_t gbcap, int64_t gpcap, int64_t ghs) {
    auto al4 = [](int64_t x) { return (x + 3) & ~3ll; };
    return 2 * al4(gbcap) + al4(gcap) + al4(gpcap) + al4(gcap) + ghs;
}

int64_t ugs_ord_words(int stages) { return (int64_t)ord_words_before(stages); }   // words holding the orders of stages [0, stages)

uint32_t ugs_chain_value(int idx) { return (idx >= 0 && idx < kChainLen) ? kChainHost[idx] : 0u; }

uint32_t ugs_chain_at_least(int64_t c, int *index_out) {   // smallest chain value >= c
    for (int i = 0; i < kChainLen; ++i) if ((int64_t)kChainHost[i] >= c) { if (index_out) *index_out = i; return kChainHost[i]; }
    if (index_out) *index_out = -1;
    return 0;
}

template <int GS, int CAP, int BLOCK>
static hipError_t launch_lds(const UgsWalkArgs &a, int cus, int blocks_per_cu, hipStream_t s, UgsLaunchInfo *info, const char *name) {
    constexpr bool kCanPad = GS == 64;      // (padded rows for the 8- and 16-lane tiers were measured in round 4: no gain on the PROTEINS- and MUTAG-shaped batches, +5 us on the QM9-shaped one)
    constexpr int GROUPS = BLOCK / GS;
    const int64_t work = a.in_list ? (int64_t)cus * blocks_per_cu * GROUPS : a.row_count;   // list length unknown on the host
    int64_t grid = (work + GROUPS - 1) / GROUPS;
    const int64_t cap = (int64_t)cus * blocks_per_cu;
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    if (kCanPad && a.plan.prow) hipLaunchKernelGGL((ugs_walk_lds<GS, CAP, BLOCK, kCanPad>), dim3((unsigned)grid), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL((ugs_walk_lds<GS, CAP, BLOCK, false>), dim3((unsigned)grid), dim3(BLOCK), 0, s, a);
    if (info) { info->name = name; info->grid = (int)grid; info->block = BLOCK; info->lds_bytes = GROUPS * TierCfg<CAP>::WORDS * 4; }
    return hipGetLastError();
}

#ifndef UGS_BLOCKS_M
#define UGS_BLOCKS_M 20
#endif
#ifndef UGS_BLOCKS_S
#define UGS_BLOCKS_S 3      // 51.7 KB of LDS and 145 VGPRs per 256-thread block: three fit a CU (C4, 65 536 rows: walk 58 -> 47 us against two)
#endif
hipError_t ugs_launch_walk(const UgsWalkArgs &a, int tier, int cus, int share_percent, hipStream_t s, UgsLaunchInfo *info) {
    if (cus <= 0) cus = 256;
    // share_percent < 100: the persistent grid takes only that share of the blocks a CU can hold, so that other kernels (the
    // collation, RCCL) find registers, LDS and wave slots on every CU while a walk is running (a full grid holds them to its end)
    auto part = [&](int blocks) { const int b = (int)((long long)blocks * (share_percent <= 0 || share_percent > 100 ? 100 : share_percent) / 100); return b < 1 ? 1 : b; };
    switch (tier) {
    case UGS_TIER_S:
        // graphs of at most 33 vertices (QM9-, MUTAG-sized): 32 candidates per walk, 27 KB of LDS and 95 VGPRs per block -- five blocks
        // per CU instead of three (the QM9-shaped batch of 65 536 rows takes two trips instead of three)
        if (a.pad == (UGS_SMALL_CAP | UGS_WIDE_LANES) && !a.in_list) return launch_lds<UGS_WIDE_LANES, UGS_SMALL_CAP, 256>(a, cus, part(5), s, info, "ugs_walk_lds<16,32>");
        if (a.pad == UGS_SMALL_CAP && !a.in_list) return launch_lds<8, UGS_SMALL_CAP, 256>(a, cus, part(5), s, info, "ugs_walk_lds<8,32>");
        // 16 lanes per walk where the host asks for them: a batch whose walks are all resident at once is bound by ONE walk's
        // latency -- four walks per wave diverge less than eight and a stage has half the elements per lane (PROTEINS-shaped batch of
        // 8192 rows: 26.7 -> 23.1 us; 32 lanes: 30.0)
        if (a.pad == UGS_WIDE_LANES && !a.in_list) return launch_lds<UGS_WIDE_LANES, 64, 256>(a, cus, part(UGS_BLOCKS_S), s, info, "ugs_walk_lds<16,64>");
        return launch_lds<8, 64, 256>(a, cus, part(UGS_BLOCKS_S), s, info, "ugs_walk_lds<8,64>");
    // one walk per wave: two walks per wave (GS 32) measured 26.4 ms vs 18.7 ms per 1M walks on C5 (two chunks per row)
    // resident one-wave blocks per CU: LDS is granted in 1280-byte granules (128 per CU) -- 7648 B = 6 granules -> 21 blocks, of
    // which the register budget (96 VGPRs: 5 waves per SIMD) admits 20; 19.5 KB = 16 granules -> 8; 38.9 KB = 31 granules -> 4
    case UGS_TIER_M: return launch_lds<64, 448, 64>(a, cus, part(UGS_BLOCKS_M), s, info, "ugs_walk_lds<64,448>");
    // 704 candidates: 11.6 KB = 10 granules -> 12 blocks per CU (3 waves per SIMD: 168 VGPRs)
    case UGS_TIER_W: return launch_lds<64, 704, 64>(a, cus, part(12), s, info, "ugs_walk_lds<64,704>");
    case UGS_TIER_X: return launch_lds<64, 1024, 64>(a, cus, part(8), s, info, "ugs_walk_lds<64,1024>");
    // 1408 candidates: 23.2 KB = 19 granules -> 6 blocks per CU (2 waves per SIMD: 256 VGPRs)
    case UGS_TIER_V: return launch_lds<64, 1408, 64>(a, cus, part(6), s, info, "ugs_walk_lds<64,1408>");
    case UGS_TIER_L: return launch_lds<64, 2048, 64>(a, cus, part(4), s, info, "ugs_walk_lds<64,2048>");
    default: {
        int64_t grid = a.gws_words_per_group > 0 ? a.gws_groups : 0;
        if (grid < 1) return hipErrorInvalidValue;
        hipLaunchKernelGGL(ugs_walk_global, dim3((unsigned)grid), dim3(64), 0, s, a);
        if (info) { info->name = "ugs_walk_global"; info->grid = (int)grid; info->block = 64; info->lds_bytes = UGS_KMAX * 4; }
        return hipGetLastError();
    }
    }
}

hipError_t ugs_launch_build_prow(const UgsPlanDev &plan, int64_t num_vertices, int2 *prow, int shift, int cus, hipStream_t s) {
    (void)cus;
    const int64_t total = num_vertices << shift;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(ugs_build_prow, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, plan, num_vertices, prow, shift);
    return hipGetLastError();
}

hipError_t ugs_launch_scan(const uint32_t *counts, int64_t rows, int64_t *edge_ptr, int64_t *block_tmp, hipStream_t s, int64_t *h_total,
                           uint32_t *h_flag, uint32_t epoch) {
    if (rows <= 0) { return hipMemsetAsync(edge_ptr, 0, sizeof(int64_t), s); }          // (callers do not ask for the signal then)
    const ScanSignal sg{h_total, h_flag, epoch};
    // one block pays only while it needs a round or two (measured: 65 536 rows in 8 rounds 60 us against 13 us + gaps for the
    // three launches; 8 192 rows in one round instead of four 256-thread rounds: C3 step 80 -> 79 us)
    if (rows <= (int64_t)2 * kScanWide * kScanPer) {
        hipLaunchKernelGGL(ugs_scan_small, dim3(1), dim3(kScanWide), 0, s, counts, rows, edge_ptr, sg);
        return hipGetLastError();
    }
    const int64_t nb = (rows + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(ugs_scan_partials, dim3((unsigned)nb), dim3(kScanBlock), 0, s, counts, rows, block_tmp);
    if (nb <= 4096) {
        hipLaunchKernelGGL(ugs_scan_final_sum, dim3((unsigned)nb), dim3(kScanBlock), 0, s, counts, rows, (const int64_t *)block_tmp, edge_ptr, sg);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(ugs_scan_block_sums, dim3(1), dim3(kScanBlock), 0, s, block_tmp, nb);
    hipLaunchKernelGGL(ugs_scan_final, dim3((unsigned)nb), dim3(kScanBlock), 0, s, counts, rows, (const int64_t *)block_tmp, edge_ptr, sg);
    return hipGetLastError();
}

// Rows staged by the walk: the items are already in output order; 16 lanes per row apply the endpoint numbering of the mode
// (reference src/sampler.cpp:258-281) and write the three output arrays.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void ugs_fill_staged(UgsFillArgs a) {
    constexpr int GS = 16, GROUPS = BLOCK / GS;
    const int lane = (int)threadIdx.x & (GS - 1), gib = (int)threadIdx.x / GS;
    const UgsPlanDev &P = a.plan;
    const int k = a.k;
    const int64_t ngroups = (int64_t)gridDim.x * GROUPS;
    for (int64_t row_rel = (int64_t)blockIdx.x * GROUPS + gib; row_rel < a.row_count; row_rel += ngroups) {
        if (!(a.counts[row_rel] & UGS_COUNT_STAGED)) continue;
        const int64_t e0 = a.edge_ptr[row_rel];
        const int n = (int)(a.edge_ptr[row_rel + 1] - e0);
        const uint2 *items = a.stage + row_rel * UGS_STAGE_ITEMS;
        const int64_t *nrow = a.nodes + row_rel * k;
        int64_t i = 0;
        const int64_t row = a.row_begin + row_rel;
        if (a.mode == 1) i = (P.num_graphs == 1) ? row : row % a.m;
        for (int t = lane; t < n && t < UGS_STAGE_ITEMS && e0 + t < a.ld; t += GS) {   // ld is also the buffers' capacity
            const uint2 x = items[t];
            const int j = (int)(x.y & 0xFFu), l = (int)(x.y >> 8);
            int64_t uf, vf;
            if (a.mode == 0) { uf = j; vf = l; }
            else if (a.mode == 1) { uf = i * k + j; vf = i * k + l; }
            else if (a.mode == 3) { uf = row_rel * k + j; vf = row_rel * k + l; }
            else { uf = nrow[j]; vf = nrow[l]; }
            a.edge_index[e0 + t] = uf;
            a.edge_index[a.ld + e0 + t] = vf;
            a.edge_src[e0 + t] = (int64_t)(int32_t)x.x;
        }
    }
}

int64_t ugs_fill_scan_tiles(int64_t rows) { return (rows + 31) / 32; }

hipError_t ugs_launch_fill_scan(const UgsFillArgs &a, int cus, hipStream_t s, UgsLaunchInfo *info) {
    if (cus <= 0) cus = 256;
    constexpr int BLOCK = 256, GROUPS = 32;
    const int64_t tiles = (a.row_count + GROUPS - 1) / GROUPS;
    int64_t grid = tiles;
    if (grid > (int64_t)cus * 8) grid = (int64_t)cus * 8;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((ugs_fill_scan<BLOCK>), dim3((unsigned)grid), dim3(BLOCK), 0, s, a);
    if (info) { info->name = "ugs_fill_scan<8>"; info->grid = (int)grid; info->block = BLOCK; info->lds_bytes = GROUPS * UGS_KMAX * 4; }
    return hipGetLastError();
}

hipError_t ugs_launch_fill(const UgsFillArgs &a, int wide, int cus, hipStream_t s, UgsLaunchInfo *info) {
    if (a.row_count <= 0) return hipSuccess;
    if (cus <= 0) cus = 256;
    if (a.stage) {
        constexpr int BLOCK = 256, GROUPS = 16;
        int64_t grid = (a.row_count + GROUPS - 1) / GROUPS;
        if (grid > (int64_t)cus * 32) grid = (int64_t)cus * 32;
        hipLaunchKernelGGL((ugs_fill_staged<BLOCK>), dim3((unsigned)grid), dim3(BLOCK), 0, s, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (wide) {
        constexpr int BLOCK = 256, GROUPS = 4;
        int64_t grid = (a.row_count + GROUPS - 1) / GROUPS;
        if (grid > (int64_t)cus * 8) grid = (int64_t)cus * 8;
        if (a.ulist && grid > cus) grid = cus;                 // leftovers of a staged call: few rows, length known on the device only
        hipLaunchKernelGGL((ugs_fill<64, BLOCK>), dim3((unsigned)grid), dim3(BLOCK), 0, s, a);
        if (info) { info->name = "ugs_fill<64>"; info->grid = (int)grid; info->block = BLOCK; info->lds_bytes = GROUPS * UGS_KMAX * 20; }
    } else {
        constexpr int BLOCK = 256, GROUPS = 32;
        int64_t grid = (a.row_count + GROUPS - 1) / GROUPS;
        if (grid > (int64_t)cus * 8) grid = (int64_t)cus * 8;
        hipLaunchKernelGGL((ugs_fill<8, BLOCK>), dim3((unsigned)grid), dim3(BLOCK), 0, s, a);
        if (info) { info->name = "ugs_fill<8>"; info->grid = (int)grid; info->block = BLOCK; info->lds_bytes = GROUPS * UGS_KMAX * 4; }
    }
    return hipGetLastError();
}
