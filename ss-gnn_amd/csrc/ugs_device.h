// ugs_device.h -- structures shared by the host library (ugs_host.cpp) and the gfx950 kernels (ugs_kernels.hip).
//
// HBM layout of a *plan* (one PyG batch, or one graph of the handle API), all arrays resident in device memory:
//
//   graphs  UgsGraphDesc[G]          one 48-byte descriptor per graph
//   rowptr  int64[sum(n_g + 1)]      ABSOLUTE offsets into adj/ecol (row r of graph g at rowptr[rbase_g + r])
//   adj     int2[nnz]                (w, rank(w)) per CSR entry, CSR order = reference build_csr order
//                                    (reference src/preproc.cpp:32-86).  rank(w) = index_of[w] is stored NEXT TO
//                                    the neighbour so the suffix filter `index_of[w] >= root_vi`
//                                    (reference src/sampler.cpp:62) costs no second (random) gather.
//   adjf    int2[nnz]                (w, ecol): the fill kernel's view of the same CSR -- neighbour and the value written to
//                                    edge_src for this entry (batch: column of the batch edge_index; handle API: column
//                                    of the graph's edge_index) side by side, so the edge column costs no extra gather
//   roots   UgsRootRec[sum(n_g)]     alias table row + both candidate root vertices in ONE 24-byte record, so the
//                                    root draw (reference include/sampler.hpp:72-77 + src/sampler.cpp:165-173) is one gather
//   viable  int2[...]                (vi, order[vi]) lists for relaxation levels 1/2 (reference src/sampler.cpp:121-150)
//   prow    int2[sum(n_g) << s]      PADDED ROWS for the one-walk-per-wave tiers (built on the device at the first such walk):
//                                    vertex v of graph g owns the 2^s entries at (vbase_g + v) << s -- entry 0 is the row's header
//                                    (CSR degree, absolute CSR position of the row's first entry), entries 1.. are the first
//                                    2^s - 1 (w, rank(w)) pairs of the row; longer rows continue in adj[].  A walk step then needs
//                                    ONE dependent memory round trip (the row, at an address computed from the vertex) instead
//                                    of two (row pointer, then row), and no line is fetched for a row-pointer pair.  Only the lines
//                                    a typical visited row fills are fetched with the header (prow_first entries); the block's
//                                    remaining lines follow for the rows that reach into them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UGS_KMAX 32          // largest supported k (per-walk vertex list lives in LDS / registers)

struct UgsGraphDesc {
    int64_t node_lo;      // added to local vertex ids in the nodes output (batch: ptr[g])
    int64_t rbase;        // first rowptr entry of this graph
    int64_t vbase;        // first roots[] entry of this graph
    int64_t viable_base;  // first viable[] entry (levels 1, 2)
    int32_t n;            // vertices
    int32_t level;        // 0: alias-weighted roots; 1, 2: uniform over viable list; -1: degenerate (n <= 0 or n < k)
    int32_t n_viable;
    int32_t pad;
};

struct UgsRootRec {
    double prob;          // alias-table acceptance probability of order position vi
    int32_t alias;        // alias order position
    int32_t v_self;       // order[vi]
    int32_t v_alias;      // order[alias]
    int32_t pad;
};

struct UgsPlanDev {
    const UgsGraphDesc *graphs;
    const int64_t *rowptr;
    const int2 *adj;
    const int2 *adjf;
    const UgsRootRec *roots;
    const int2 *viable;
    int64_t num_graphs;
    const int2 *prow;        // padded rows (NULL until built; the 8-lane and global-memory tiers use rowptr + adj)
    int32_t prow_shift;      // log2(entries per padded row), 3..6
    int32_t prow_first;      // entries fetched before the degree is known (whole 128-byte lines); the rest of the block only if the row needs it
};

struct UgsWalkArgs {
    UgsPlanDev plan;
    int32_t m;               // samples per graph
    int32_t k;
    int32_t mode;            // UGS_MODE_* / UGS_EDGE_* (0 local, 1 flat, 2 global)
    int32_t pad;             // UGS_SMALL_CAP: first-tier launch of tier S in its 32-candidate form; UGS_WIDE_LANES: tier S with 16 lanes per walk (the two may be ORed)
    int64_t extra_node_off;  // handle API "global": base_offset
    uint64_t seed64;         // (uint64_t)(int64_t)seed
    const uint64_t *seed_ptr;// if not NULL the seed is read from here (captured HIP graphs: the value changes between replays)
    int64_t row_begin;       // first of the G*m rows produced by this call
    int64_t row_count;
    int64_t *nodes;          // [row_count, k]
    uint32_t *counts;        // [row_count] edge entries per row; top bit (UGS_COUNT_STAGED) = the row's items are in `stage`
    // overflow hand-off between tiers: rows whose candidate set outgrew the tier's LDS capacity
    const int64_t *in_list;  // NULL: process rows 0..row_count-1; else process in_list[0..*in_count)
    const uint32_t *in_count;
    int64_t *ovf_list;       // rows (relative to row_begin) handed to the next tier
    uint32_t *ovf_count;
    // global-memory workspace of the last tier (per-group slices)
    uint32_t *gws;
    int64_t gws_words_per_group;
    int64_t gws_groups;      // number of workspace slices = grid of the global tier
    int32_t gcap;            // candidate capacity of the global tier
    int32_t ghs;             // hash slots (power of two) of the global tier
    int32_t gbcap;           // bucket-table entries of the global tier
    int32_t gpcap;           // words of all materialised stage orders of the global tier
    // induced edges staged by the walk itself (one-walk-per-wave LDS tiers; NULL = off): the walk meets every induced edge
    // when it scans the row of the later endpoint, so complete rows leave their directed edge items here in OUTPUT order
    // and the fill kernel only expands them.  Rows that do not fit (or ran in a tier without staging) are flagged 0 and,
    // if they have edges, listed for the row-reading fill kernel.
    // dynamic work distribution: the first gridsize*groups items are taken statically, every further item index is
    // groups_total + atomicAdd(work_next, 1) (NULL: static striding).  A walk's cost varies (degrees, which stages of the order
    // get invalidated), so a static split ends with the unluckiest wave.
    unsigned long long *work_next;
    uint32_t *wsum;          // 8-lane tier, optional: [ceil(row_count / 8)] sum of the counts of every 8 consecutive rows (for ugs_fill_scan)
    uint2 *stage;            // [row_count, UGS_STAGE_ITEMS]: x = batch column, y = source local index | target local index << 8
    int64_t *ulist;          // rows (relative) with edges that are NOT staged
    uint32_t *ucount;
};
#define UGS_COUNT_STAGED 0x80000000u
#define UGS_STAGE_ENTRIES 32    /* undirected hits a walk can hold in LDS */
#define UGS_STAGE_ITEMS 64      /* directed items per row in the staging buffer */

struct UgsFillArgs {
    UgsPlanDev plan;
    int32_t m, k, mode, pad;
    int64_t extra_node_off;
    int64_t row_begin, row_count;
    const int64_t *nodes;
    const int64_t *edge_ptr;   // [row_count + 1]
    int64_t *edge_index;       // [2, ld]
    int64_t ld;
    int64_t *edge_src;
    const uint2 *stage;        // staging left by the walk of the same rows (NULL: every row is filled from its adjacency rows)
    const uint32_t *counts;    // the walk's per-row counts: their top bit says whether the row was staged
    const int64_t *ulist;      // with staging: the rows the row-reading kernel still has to do
    const uint32_t *ucount;
    // scan folded into the fill (small-batch step, ugs_fill_scan): the kernel turns the walk's per-row counts into edge_ptr itself --
    // tiles of 32 rows, a tile's offset from the sums of 8 rows the walk kernel left
    int64_t *edge_ptr_out;              // [row_count + 1], written by the kernel (NULL: edge_ptr above is read)
    const uint32_t *wsum;               // [ceil(row_count / 8)]: UgsWalkArgs::wsum of the walk of the same rows
    // ugs_fill_scan for the library's own jobs (the drop-in call): the edge buffers are ONE staging area of `packed_cap` int64 words and
    // the kernel lays the outputs out for the total it computes itself -- edge_index [2, total] and edge_src [total] behind it, what
    // the caller's tensors look like, so that they leave in one copy -- or writes nothing if 3 * total exceeds the capacity (the host
    // sees the total and fills the ordinary way).  The block that owns the last tile hands the total to the host as soon as it
    // knows it (pinned words, epoch protocol of the scan kernels), before it fills its rows.
    int64_t packed_cap;                 // 0: off (edge_index / ld / edge_src as given)
    int64_t *h_total;
    uint32_t *h_flag;
    uint32_t epoch;
};

struct UgsLaunchInfo {
    const char *name;
    int grid, block, lds_bytes;
};

// tiers of the walk kernel: candidate-set capacity held in LDS per walk, lanes per walk
enum { UGS_TIER_S = 0 /* cap 64, 8 lanes */, UGS_TIER_M = 1 /* cap 448, 64 lanes */, UGS_TIER_W = 2 /* cap 704, 64 lanes, half-size bucket table */,
       UGS_TIER_X = 3 /* cap 1024, 64 lanes */, UGS_TIER_V = 4 /* cap 1408, 64 lanes, third-size bucket table */, UGS_TIER_L = 5 /* cap 2048, 64 lanes */,
       UGS_TIER_G = 6 /* global-memory workspace, 64 lanes */ };
#define UGS_LDS_TIERS 6
static constexpr int UGS_TIER_CAP[UGS_LDS_TIERS] = {64, 448, 704, 1024, 1408, 2048};
static constexpr int UGS_TIER_LANES[UGS_LDS_TIERS] = {8, 64, 64, 64, 64, 64};                // lanes per walk
static constexpr int UGS_TIER_HASH_LIMIT[UGS_LDS_TIERS] = {96, 448, 896, 1536, 1792, 3072};  // TierCfg<CAP>::HLIMIT (static_assert in ugs_kernels.hip)
// a form of tier S with half the workspace, for plans whose walks cannot hold more than 32 candidates (UgsWalkArgs::pad = UGS_SMALL_CAP)
#define UGS_SMALL_CAP 32
// tier S with 16 lanes per walk instead of 8 (UgsWalkArgs::pad = UGS_WIDE_LANES): first-tier launches whose walks are all resident at
// once (row_count <= CUs x blocks per CU x 16) of plans whose walks cannot be handed on -- such a launch lasts as long as one walk
#define UGS_WIDE_LANES 16
#define UGS_SMALL_HASH_LIMIT 48

hipError_t ugs_launch_walk(const UgsWalkArgs &a, int tier, int device_cus, int share_percent, hipStream_t s, UgsLaunchInfo *info);
hipError_t ugs_launch_build_prow(const UgsPlanDev &plan, int64_t num_vertices, int2 *prow, int shift, int device_cus, hipStream_t s);
#define UGS_COLLATE_MAX_WORLD 64
hipError_t ugs_launch_collate_unpack(const void *d_msgs, int world, int64_t msg_bytes, const int64_t *row_off, int k, int node_b, int eidx_b,
                                     int esrc_b, int64_t rows_cap, int64_t edge_cap, const int64_t *section_off4, int64_t *d_nodes,
                                     int64_t *d_edge_index, int64_t ld, int64_t *d_edge_ptr, int64_t *d_edge_src, int64_t *d_max_total,
                                     hipStream_t s);
// out[i] = in[i] + base, i < n (a row chunk's edge_ptr moved to its place in the whole call's: ugs_sample_batch_stream)
hipError_t ugs_launch_rebase_edge_ptr(const int64_t *in, int64_t *out, int64_t n, int64_t base, hipStream_t s);
// device batch pass (ugs_batch.hip): slicing, LRU keys and CSR of a batch of small graphs; limits per graph of that path
#define UGS_BATCH_PASS_MAX_COLS 1000   /* a key covers every column up to here (reference include/cache.hpp:100 samples longer graphs) */
#define UGS_BATCH_PASS_MAX_N 2048
#define UGS_BATCH_PASS_FUSED_WORK (4ll << 20)   /* G * E up to here: the build kernel slices the batch itself (one launch) */
int64_t ugs_batch_pass_fused_work();             /* the limit in force: UGS_BP_FUSED_WORK overrides it (testing aid: 0 = always two kernels) */
hipError_t ugs_launch_batch_pass(const int64_t *d_src, const int64_t *d_dst, int64_t E, const int64_t *d_ptr, int64_t G, int k,
                                 int32_t *d_owner, uint32_t *d_cnt_jminc_jmax, const int64_t *d_rstart, int64_t *d_rowptr, int2 *d_adj,
                                 int2 *d_adjf, int32_t *d_vrank, unsigned long long *d_bump, unsigned long long bump_base, uint32_t epoch, void *h_back,
                                 unsigned long long *d_done, unsigned long long done_base, hipStream_t s);
// graphs of a device-built plan the LRU does not know (cold path): the rest of their preprocessing on the device (ugs_bp_roots)
#define UGS_BATCH_ROOTS_MAX_N 1024     /* larger unknown graphs are preprocessed on the host, as before */
struct UgsBpMissIn { int32_t g; int32_t pad; int64_t roots_off; int64_t via_off; };           // arena offsets (elements) reserved by the host
struct UgsBpMissOut { int32_t level, n_viable, nonzero, max_deg; double Z, sb_deg; };
hipError_t ugs_launch_batch_roots(const int64_t *d_ptr, const int64_t *d_rstart, const int64_t *d_rowptr, const int2 *d_adj, const int32_t *d_vrank,
                                  const UgsBpMissIn *h_in, UgsBpMissOut *h_out, int64_t misses, int k, UgsRootRec *d_roots, int2 *d_via,
                                  unsigned long long *d_done, unsigned long long done_base, uint32_t *h_done, uint32_t epoch, hipStream_t s);
// h_total / h_flag (pinned host memory, or NULL): the kernel also hands the total to the host and signals with `epoch`
hipError_t ugs_launch_scan(const uint32_t *counts, int64_t rows, int64_t *edge_ptr, int64_t *block_tmp, hipStream_t s, int64_t *h_total = nullptr,
                           uint32_t *h_flag = nullptr, uint32_t epoch = 0);
hipError_t ugs_launch_fill(const UgsFillArgs &a, int wide, int device_cus, hipStream_t s, UgsLaunchInfo *info);
// scan + fill in one launch (8-lane tier, rows read from their adjacency)
hipError_t ugs_launch_fill_scan(const UgsFillArgs &a, int device_cus, hipStream_t s, UgsLaunchInfo *info);
int64_t ugs_fill_scan_tiles(int64_t rows);
int64_t ugs_scan_tmp_words(int64_t rows);
int64_t ugs_global_ws_words(int64_t gcap, int64_t gbcap, int64_t gpcap, int64_t ghs);
uint32_t ugs_chain_at_least(int64_t c, int *index_out);
uint32_t ugs_chain_value(int idx);
int64_t ugs_ord_words(int stages);
