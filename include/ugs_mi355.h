/* ugs_mi355.h -- C ABI of libugs_mi355.so, the MI355X-native (gfx950 / HIP) uniform k-subgraph sampler.
 *
 * This is the drop-in boundary for the reference's `ugs_sampler` plugin (AniruddhaMandal/SS-GNN,
 * src/samplers/ugs_sampler).  The reference exposes a pybind11 module (src/extension.cpp:4-13) taking
 * torch::Tensor arguments; this library exposes the same operations over plain pointers and sizes so
 * that any host language can bind them (ctypes stub: ss-gnn_amd/ugs_sampler/__init__.py; see
 * INTEGRATION.md).  No torch types appear in any signature.
 *
 * Conventions
 *   - every function returns 0 on success or a negative UGS_E_* code; ugs_last_error() returns the
 *     message of the calling thread's last failure (same text as the reference's exception where the
 *     reference has one).
 *   - `edge_index` is int64 [2, E] with explicit row stride (elements) so non-contiguous tensors need no copy:
 *     source of column j = edge_index[j], destination = edge_index[row_stride + j].
 *   - all outputs are int64, caller-allocated; `dst_is_device` != 0 means the output pointers are
 *     device (HBM) pointers, otherwise host pointers (pinned or pageable).
 *   - sampling always runs on the GPU: there is no CPU fallback.  Without a usable HIP device every
 *     sampling entry point fails with UGS_E_NO_DEVICE.
 */
#ifndef UGS_MI355_H
#define UGS_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UGS_OK 0
#define UGS_E_INVALID_HANDLE (-1)   /* "Invalid preproc handle"            reference src/sampler.cpp:105 */
#define UGS_E_NO_ROOTS (-2)         /* "No viable roots available"         reference src/sampler.cpp:149 */
#define UGS_E_BAD_MODE (-3)         /* "mode must be one of: ..."          reference src/ugs_sampler_batch_extension.cpp:89-90 */
#define UGS_E_BAD_ARG (-4)
#define UGS_E_NO_DEVICE (-5)        /* no usable HIP device / kernel image */
#define UGS_E_HIP (-6)              /* a HIP runtime call failed */
#define UGS_E_UNSUPPORTED (-7)      /* outside the limits documented in DESIGN.md (k > 32, nnz >= 2^31 per plan, ...) */
#define UGS_E_EDGE_SRC (-8)         /* edge_src range check                reference src/ugs_sampler_batch_extension.cpp:213-222 */
#define UGS_E_CAPACITY (-9)         /* ugs_sample_batch_stream: the call produced more edge entries than edge_capacity */

/* edge_mode of sample()            reference src/sampler.cpp:95 ("local" | "flat" | "global") */
#define UGS_EDGE_LOCAL 0
#define UGS_EDGE_FLAT 1
#define UGS_EDGE_GLOBAL 2
/* mode of sample_batch()           reference src/ugs_sampler_batch_extension.cpp:81 ("sample" | "graph" | "global") */
#define UGS_MODE_SAMPLE 0
#define UGS_MODE_GRAPH 1
#define UGS_MODE_GLOBAL 2
/* ugs_plan_fill only (no reference counterpart): endpoints numbered r * k + local index, r = row's position inside this call's row range --
 * the edge index the consumer builds from edge_index_t + repeat_interleave(arange(B), counts) * k
 * (reference src/gps/gps/models/ss_gnn.py:463-464), so that step (and its host synchronisation) disappears */
#define UGS_FILL_BATCH 3

const char *ugs_last_error(void);
const char *ugs_version(void);

/* Number of usable HIP devices (0 is an error: UGS_E_NO_DEVICE). */
int ugs_device_count(int *count);
/* Select the HIP device used by the calling thread's subsequent calls (default: current HIP device). */
int ugs_set_device(int device);

/* Stream of the calling thread's subsequent JOBS (ugs_sample_*, ugs_sample_batch_*, ugs_eps_*): `use` != 0 runs their kernels
 * and copies on `stream` (a hipStream_t; NULL = the default stream) instead of the library's own non-blocking stream; `use` = 0
 * restores the library's stream.  A caller that hands in DEVICE output buffers obtained from a stream-ordered allocator
 * (torch.empty on torch's current stream) must run the job on that stream: a block the allocator just recycled may still be
 * read by kernels queued there, and only stream order keeps the job's writes behind them.  The plan API takes its stream per call. */
int ugs_set_stream(void *stream, int use);

/* ---- preprocessing handles: replaces create_preproc / destroy_preproc / has_graphlets / get_preproc_info
 *      (reference src/preproc.cpp:262-314, pybind names src/extension.cpp:7-10) -------------------------------- */
int ugs_create_preproc(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, int64_t num_nodes, int k,
                       int64_t *handle_out);
int ugs_destroy_preproc(int64_t handle);                       /* unknown handles are ignored, like the reference */
int ugs_has_graphlets(int64_t handle, int *out);               /* unknown handle -> 0 */
int ugs_get_preproc_info(int64_t handle, int *found, int64_t *num_nodes, int64_t *num_edges_stored, double *Z,
                         int *bucket_count_nonzero);
/* internals of a handle for the preprocessing parity tests (any pointer may be NULL):
 * indptr int64[n+1], indices int32[nnz], edge_col int32[nnz], order int32[n], index_of int32[n],
 * suffix_deg int32[n], bucket_b double[n], prob double[n], alias int32[n] */
int ugs_preproc_dump(int64_t handle, int64_t *indptr, int32_t *indices, int32_t *edge_col, int32_t *order,
                     int32_t *index_of, int32_t *suffix_deg, double *bucket_b, double *prob, int32_t *alias);

/* ---- sample(): replaces sample(handle, m_per_graph, k, edge_mode, base_offset, seed)
 *      (reference src/sampler.cpp:91-290).  Two phases so that the caller owns the outputs:
 *      begin  runs the walks on the GPU and reports the number of edge entries;
 *      finish writes nodes[m,k], edge_index[2,total_edges], edge_ptr[m+1], edge_src[total_edges] and frees the job.
 *      ugs_job_cancel frees a job that will not be finished. ------------------------------------------------- */
typedef struct ugs_job ugs_job;
int ugs_sample_begin(int64_t handle, int m_per_graph, int k, int edge_mode, int64_t base_offset, int seed,
                     ugs_job **job_out, int64_t *total_edges_out);
int ugs_sample_finish(ugs_job *job, int64_t *nodes, int64_t *edge_index, int64_t *edge_ptr, int64_t *edge_src,
                      int dst_is_device);

/* ---- sample_batch(): replaces sample_batch(edge_index, ptr, m_per_graph, k, mode, seed)
 *      (reference src/ugs_sampler_batch_extension.cpp:77-299), including its process-global LRU of
 *      preprocessing handles keyed by an FNV-1a hash that ignores k (include/cache.hpp:81-109).
 *      finish writes nodes[G*m,k], edge_index[2,total_edges], edge_ptr[G*m+1], sample_ptr[G+1],
 *      edge_src_global[total_edges]. ----------------------------------------------------------------------- */
int ugs_sample_batch_begin(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr,
                           int64_t num_graphs, int m_per_graph, int k, int mode, int seed, ugs_job **job_out,
                           int64_t *total_edges_out);
int ugs_sample_batch_finish(ugs_job *job, int64_t *nodes, int64_t *edge_index, int64_t *edge_ptr,
                            int64_t *sample_ptr, int64_t *edge_src_global, int dst_is_device);
int ugs_job_cancel(ugs_job *job);

/* The same call (reference src/ugs_sampler_batch_extension.cpp:77-299, same LRU, same results) for LARGE host-visible batches,
 * in one piece: the caller hands over its (pinned) host buffers up front -- nodes[G*m,k], edge_ptr[G*m+1], sample_ptr[G+1],
 * edge_src_global[edge_capacity] and edge_index_out[2*edge_capacity] -- and the rows are sampled in chunks whose results cross
 * PCIe on a second stream while the next chunk walks (the two-phase call copies only after the last walk).  On return
 * *total_edges_out = total, edge_index_out holds [2, total] CONTIGUOUSLY at its start (row 1 begins at edge_index_out + total) and
 * edge_src_global[total].  edge_capacity is the caller's estimate (e.g. the total of an earlier call on the same batch plus a
 * margin); a call that needs more returns UGS_E_CAPACITY with the buffers' contents undefined and *total_edges_out = the entries
 * reached when the room ran out (> edge_capacity, a lower bound on the total) -- repeat it through ugs_sample_batch_begin / _finish.
 * k = 1 or m_per_graph = 0 produce no entries: edge_capacity 0 with null edge buffers is accepted.  UGS_STREAM_CHUNK_ROWS overrides the chunk size (default: an eighth of the rows, >= 65536). */
int ugs_sample_batch_stream(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr,
                            int64_t num_graphs, int m_per_graph, int k, int mode, int seed, int64_t edge_capacity,
                            int64_t *nodes, int64_t *edge_index_out, int64_t *edge_ptr, int64_t *sample_ptr,
                            int64_t *edge_src_global, int64_t *total_edges_out);
/* sample() of the handle API (reference src/sampler.cpp:91-290) streamed the same way: nodes[m,k], edge_ptr[m+1],
 * edge_src[edge_capacity], edge_index_out[2*edge_capacity] (holds [2, total] contiguously on return); UGS_E_CAPACITY as above. */
int ugs_sample_stream(int64_t handle, int m_per_graph, int k, int edge_mode, int64_t base_offset, int seed,
                      int64_t edge_capacity, int64_t *nodes, int64_t *edge_index_out, int64_t *edge_ptr, int64_t *edge_src,
                      int64_t *total_edges_out);
/* ugs_sample_batch_stream starts early: a batch whose 32 sampled words (first / last / evenly spaced columns, ptr ends) match a
 * batch seen before begins its walks on that batch's plan while the real lookup -- the content hash over every column and the LRU
 * replay of include/cache.hpp:81-109 -- runs on a helper thread; the results stand only if the lookup names the same plan, otherwise
 * the streams are drained and the call runs again on the right plan (same results, the early work is lost).  Counters since process
 * start: early starts that stood / that were thrown away.  UGS_NO_SPECULATION set = always look up first.  ugs_sample_batch_begin does
 * the same for batches of >= 2^21 columns (UGS_SPEC_MIN_COLS overrides the threshold: testing aid) and counts here too. */
int ugs_stream_stats(int64_t *early_starts_kept, int64_t *early_starts_discarded);

/* LRU of preprocessing handles used by ugs_sample_batch_* (capacity from UGS_CACHE_SIZE, default 1000;
 * reference src/ugs_sampler_batch_extension.cpp:15-38).  Clearing it is the equivalent of a fresh process. */
int ugs_cache_clear(void);
int ugs_cache_stats(int64_t *size, int64_t *hits, int64_t *misses);
/* The input side of ugs_sample_batch_begin / ugs_plan_create_batch for batches of small graphs runs on the device (SURVEY.md
 * 8(f) N4): one pass over edge_index + ptr replaces the reference's per-graph slicing (src/ugs_sampler_batch_extension.cpp:41-75),
 * hashing (include/cache.hpp:81-109) and CSR construction (src/preproc.cpp:32-86); the host replays the LRU on the G keys that
 * come back.  Counters since process start: plans built that way / calls that took the general (host) path instead
 * (UGS_DEVICE_BATCH=0, non-monotone ptr, a graph with more than 2048 vertices or 1000 columns). */
int ugs_batch_pass_stats(int64_t *device_plans, int64_t *general_path);

/* ---- device-resident plans: the batch (or single graph) preprocessed once and kept in HBM, sampled many times,
 *      optionally over a sub-range of the G*m result rows (multi-GPU sharding: row b = g*m + i depends only on
 *      (graph g, seed, i)).  All pointers passed to walk/fill are DEVICE pointers; `stream` is a hipStream_t
 *      (NULL = default stream).  Nothing here synchronises with the host unless stated. ------------------------ */
typedef struct ugs_plan ugs_plan;
/* Builds (or fetches from the plan cache) the plan of a batch through the same LRU as ugs_sample_batch_begin. */
int ugs_plan_create_batch(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr,
                          int64_t num_graphs, int k, ugs_plan **plan_out);
/* Plan of one preprocessing handle (the handle API's graph). */
int ugs_plan_create_handle(int64_t handle, ugs_plan **plan_out);
int ugs_plan_release(ugs_plan *plan);
/* What the walk kernels read for graph `graph` of a plan, copied back from HBM (parity tests of the preprocessing that runs on the
 * device, incl. the cold path of the device batch pass: graphs the LRU does not know get their root records from ugs_bp_roots;
 * the host path's counterpart is ugs_preproc_dump).  level 0: prob / alias / v_self = order[vi] / v_alias = order[alias[vi]] per
 * order position (reference include/sampler.hpp:44-69 + src/preproc.cpp:176-256); levels 1, 2: the viable list (vi, order[vi])
 * (src/sampler.cpp:121-150).  Arrays hold `capacity` entries; any pointer may be NULL.  Synchronises with the device. */
int ugs_plan_graph_roots(ugs_plan *plan, int64_t graph, int64_t capacity, int32_t *level, int32_t *num_nodes, int32_t *num_viable,
                         double *prob, int32_t *alias, int32_t *v_self, int32_t *v_alias, int32_t *viable_vi, int32_t *viable_v);
/* A second plan over the same device arrays with private scratch (a plan's scratch serves one stream at a time): two steps in
 * flight on two streams go through a plan and its twin alternately.  Release both; the arrays live until the last one goes. */
int ugs_plan_twin(ugs_plan *plan, int k, ugs_plan **twin_out);
/* num_graphs, total vertices, total CSR entries, bytes resident in HBM, walk-kernel tier chosen for k */
int ugs_plan_info(const ugs_plan *plan, int k, int64_t *num_graphs, int64_t *num_vertices, int64_t *nnz,
                  int64_t *device_bytes, int *tier);
/* Walk phase for rows [row_begin, row_begin+row_count) of the G*m rows: writes d_nodes[row_count,k] and
 * d_edge_ptr[row_count+1] (exclusive scan of the per-row edge-entry counts, starting at 0).
 * If total_edges_host is not NULL the stream is synchronised and the total is returned there. */
int ugs_plan_walk(ugs_plan *plan, int m_per_graph, int k, int mode, int64_t extra_node_offset, int seed,
                  int64_t row_begin, int64_t row_count, void *stream, int64_t *d_nodes, int64_t *d_edge_ptr,
                  int64_t *total_edges_host);
/* Fill phase: writes d_edge_index[2, ld] and d_edge_src[ld] for the same rows.  ld is the row stride of d_edge_index AND the
 * capacity (in edge entries) of both buffers: entries at positions >= ld are not written (a caller that sized the buffers from
 * an estimate compares d_edge_ptr[row_count] with ld afterwards). */
int ugs_plan_fill(ugs_plan *plan, int m_per_graph, int k, int mode, int64_t extra_node_offset, int64_t row_begin,
                  int64_t row_count, void *stream, const int64_t *d_nodes, const int64_t *d_edge_ptr,
                  int64_t *d_edge_index, int64_t ld, int64_t *d_edge_src);
/* Walk + fill of the same rows as ONE call, for callers that hand over edge buffers of capacity ld up front (no host read-back
 * of the total in between: d_edge_ptr[row_count] holds it afterwards).  Same outputs as ugs_plan_walk followed by ugs_plan_fill
 * (reference src/sampler.cpp:91-290).  Knowing that nobody reads edge_ptr between the two phases, the step of a batch of small
 * graphs runs in two launches instead of three: the fill kernel scans the per-row counts itself (decoupled look-back over tiles
 * of 32 rows; `UGS_NO_FUSED_SCAN` set = the three-launch form). */
int ugs_plan_step(ugs_plan *plan, int m_per_graph, int k, int mode, int64_t extra_node_offset, int seed, int64_t row_begin,
                  int64_t row_count, void *stream, int64_t *d_nodes, int64_t *d_edge_ptr, int64_t *d_edge_index, int64_t ld,
                  int64_t *d_edge_src);
/* Share (1..100 percent, default 100) of the blocks a CU can hold that this plan's walk kernels occupy.  The walk kernels are
 * persistent grids that keep every CU's registers, LDS and wave slots to their end; a job that runs other kernels BESIDE a walk
 * (the collation of the previous batch and its RCCL transfer on another stream) lowers the share so that those find room on
 * every CU instead of queueing behind the walk. */
int ugs_plan_set_walk_share(ugs_plan *plan, int percent);
/* A whole step (seed upload, walk tiers, scan, fill) of a plan captured ONCE as a HIP graph and replayed with a new seed:
 * for batches of small graphs a step is a handful of launches for tens of microseconds of work, and the graph removes the
 * per-launch gaps.  The buffers are the caller's (d_edge_index[2, ld], d_edge_src[ld]: ld >= the largest total it expects;
 * entries beyond ld are not written, and a replay whose total exceeds ld must not be used -- compare d_edge_ptr[row_count] with ld).  No reference counterpart
 * (the reference launches nothing); same results as ugs_plan_walk + ugs_plan_fill with that seed.  Launches of one graph
 * must be issued by one thread at a time and on one stream at a time (the outputs are the graph's buffers); up to 256 replays
 * may be in flight (ring of pinned seed slots). */
typedef struct ugs_graph ugs_graph;
int ugs_plan_graph_create(ugs_plan *plan, int m_per_graph, int k, int mode, int64_t extra_node_offset, int64_t row_begin,
                          int64_t row_count, int64_t *d_nodes, int64_t *d_edge_ptr, int64_t *d_edge_index, int64_t ld,
                          int64_t *d_edge_src, ugs_graph **graph_out);
int ugs_plan_graph_launch(ugs_graph *graph, int seed, void *stream);
int ugs_plan_graph_destroy(ugs_graph *graph);
/* Name and per-launch statistics of the kernels the last ugs_plan_walk / ugs_plan_fill on this plan launched
 * (grid, block, LDS bytes) -- used by bench.py to label its roofline line. */
int ugs_plan_last_launch(const ugs_plan *plan, char *name_buf, int name_buf_len, int *grid, int *block,
                         int *lds_bytes, int64_t *overflow_rows);

/* ---- collation of a sharded batch (multi-GPU: SURVEY.md section 8(e); the reference is single-process and has no counterpart).
 *      Rank r samples the contiguous row range [row_off[r], row_off[r+1]) of the G*m rows; the finished batch is collated on
 *      the rank that feeds the trainer by ONE fixed-size message per rank (any transport: RCCL gather / all-gather of bytes).
 *      Wire format of a message (little endian, every section 16-byte aligned; ugs_collate_layout gives the offsets):
 *        header    int64 rows, int64 edge entries of this rank
 *        nodes     [rows_cap, k]  int32 | int64                     (int32 when every node id fits)
 *        edge_ptr  [rows_cap + 1] uint32, rank-local (starts at 0)
 *        edge_index [2, edge_cap] uint8 | int32 | int64              (uint8: mode "sample", local ids < k)
 *        edge_src  [edge_cap]     int32 | int64
 *      rows_cap / edge_cap are the job's fixed capacities (>= every rank's rows / edge entries), so message size does not depend
 *      on a step's outcome and no host round trip is needed per step.
 *      ugs_collate_unpack turns `world` messages (contiguous, d_msgs[world][msg_bytes]) into the batch's int64 tensors on the
 *      device: offsets are taken from the headers ON the device, entries at positions >= ld are not written.  A message whose
 *      edge total exceeds edge_cap was truncated by its sender: the batch is then INVALID; d_max_total (one device word, kept
 *      by the caller across steps) receives the largest total seen so that the caller can find out without a per-step host
 *      round trip (total > edge_cap). */
int ugs_collate_layout(int k, int node_bytes, int eidx_bytes, int esrc_bytes, int64_t rows_cap, int64_t edge_cap,
                       int64_t *section_off4, int64_t *msg_bytes);
int ugs_collate_unpack(const void *d_msgs, int world, const int64_t *row_off /* host, world+1 */, int k, int node_bytes,
                       int eidx_bytes, int esrc_bytes, int64_t rows_cap, int64_t edge_cap, int64_t *d_nodes,
                       int64_t *d_edge_index, int64_t ld, int64_t *d_edge_ptr, int64_t *d_edge_src,
                       int64_t *d_max_total /* optional: max(*d_max_total, every message's edge total), for a lazy capacity check */,
                       void *stream);

/* ---- epsilon_uniform_sampler.sample_batch(edge_index, ptr, m_per_graph, k, mode, seed, epsilon): replaces the reference's
 *      src/samplers/epsilon_uniform_sampler/src/epsilon_uniform_sampler.cpp:122-377.
 *      Random frontier growth (:18-87) with acceptance min(1, eps/(w+eps)) (:238), at most max(10, 10/eps) attempts per
 *      sample (:207); nodes of a sample sorted ascending (:256); edges = the batch columns with both endpoints in the sample,
 *      in column order, each once (:265-291); mode 0 ("sample") numbers endpoints 0..k-1 by sorted position, any other mode
 *      uses batch node ids; failed samples are rows of -1 without edges.
 *      The reference is NOT deterministic here (per-thread generators seeded with the OpenMP thread id, dynamic schedule,
 *      rows written in thread-completion order, :209-319).  This implementation is deterministic: one counter-based
 *      generator per (row, attempt), rows in graph order; parity with the reference is statistical.
 *      Same two-phase job protocol as ugs_sample_batch_*; finish writes nodes[G*m,k], edge_index[2,total],
 *      edge_ptr[G*m+1], sample_ptr[G+1], edge_src[total]. */
int ugs_eps_sample_batch_begin(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr,
                               int64_t num_graphs, int m_per_graph, int k, int mode, uint64_t seed, double epsilon,
                               ugs_job **job_out, int64_t *total_edges_out);
int ugs_eps_sample_batch_finish(ugs_job *job, int64_t *nodes, int64_t *edge_index, int64_t *edge_ptr, int64_t *sample_ptr,
                                int64_t *edge_src, int dst_is_device);

/* ---- apx_ugs_sampler.sample_batch(edge_index, ptr, m_per_graph, k, mode, seed, epsilon): replaces the reference's
 *      src/samplers/apx_ugs_sampler/src/apx_ugs_sampler.cpp:461-519.  First graph only;
 *      ptr[0]:ptr[1] is a range of edge COLUMNS (:15-33).  The reference draws everything from ONE sequential
 *      std::mt19937_64 stream, which has no parallel bit-exact form: this entry point is a HOST computation that consumes the
 *      same generator in the same order (bit-exact on the same toolchain).  It is not part of the GPU hot path.
 *      samples_out: capacity m_per_graph * k int64, sample s at samples_out[s*k .. s*k+k); *num_samples_out = S <= m_per_graph
 *      (failed samples are dropped, like the reference). */
int ugs_apx_sample_batch(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr, int64_t ptr_len,
                         int m_per_graph, int k, uint64_t seed, double epsilon, int64_t *samples_out, int64_t *num_samples_out);

/* GPU variant of the same entry point: the same algorithm with one generator per (sample, trial) -- all samples and thousands of
 * trials run side by side, a sample's result is its accepted trial with the smallest index (deterministic in (graph, seed)).
 * Parity with the reference is statistical (same output law), not bit-wise.  2 <= k <= 8.
 * order_pos_out / est_out (optional, capacity order_capacity >= number of vertices): position of every vertex in the APX-DD order
 * and its bucket estimate, for the law check of the tests. */
int ugs_apx_gpu_sample_batch(const int64_t *edge_index, int64_t row_stride, int64_t num_cols, const int64_t *ptr, int64_t ptr_len,
                             int m_per_graph, int k, uint64_t seed, double epsilon, int64_t *samples_out, int64_t *num_samples_out,
                             int32_t *order_pos_out, double *est_out, int64_t order_capacity);

/* Per-kernel timing with HIP events recorded on the launch stream (off by default).  get_timing synchronises the
 * recorded events, returns summed milliseconds and launch counts for [0] the first-tier walk kernel, [1] overflow
 * tiers + scan kernels, [2] the fill kernel since the last call, and clears them. */
int ugs_plan_set_timing(ugs_plan *plan, int on);
int ugs_plan_get_timing(ugs_plan *plan, double *ms_sum3, int64_t *launches3);

#ifdef __cplusplus
}
#endif
#endif /* UGS_MI355_H */
