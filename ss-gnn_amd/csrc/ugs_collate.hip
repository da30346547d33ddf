// ugs_collate.hip -- the one exchange step of the multi-GPU path: collation of the ranks' row ranges
// into the batch on the rank that feeds the trainer.  The reference has no counterpart (single process); what must come out
// is the reference's tensors for the whole batch (src/ugs_sampler_batch_extension.cpp:244-299): nodes [B,k], edge_index
// [2,Es], edge_ptr [B+1], edge_src [Es], int64, rows in batch order.
//
// Every rank sends ONE fixed-size message (ugs_mi355.h: wire format) over RCCL; this kernel pair turns the `world` messages
// into the final tensors in one pass: offsets come from the message headers on the device (no host round trip), narrow wire
// types are widened to int64, and each rank's rows / edge entries land at their place in batch order.
#include "ugs_device.h"

namespace {

struct CollateArgs {
    const unsigned char *msgs;     // [world, msg_bytes]
    int64_t msg_bytes;
    int world, k, node_b, eidx_b, esrc_b, pad;
    int64_t rows_cap, edge_cap;
    int64_t off_nodes, off_eptr, off_eidx, off_esrc;   // byte offsets of the sections inside a message
    int64_t row_off[UGS_COLLATE_MAX_WORLD + 1];         // first batch row of every rank (host-known: the row ranges are fixed)
    int64_t *nodes, *edge_index, *edge_ptr, *edge_src;
    int64_t ld;
    long long *max_total;                               // optional: largest per-rank edge total seen (the caller's capacity check)
};

__device__ __forceinline__ int64_t widen(const unsigned char *p, int64_t i, int bytes, bool sign) {
    switch (bytes) {
    case 1: return sign ? (int64_t)reinterpret_cast<const int8_t *>(p)[i] : (int64_t)p[i];
    case 4: return sign ? (int64_t)reinterpret_cast<const int32_t *>(p)[i] : (int64_t)reinterpret_cast<const uint32_t *>(p)[i];
    default: return reinterpret_cast<const int64_t *>(p)[i];
    }
}

__device__ __forceinline__ int64_t edge_offset(const CollateArgs &a, int r) {      // edge entries of the ranks before r
    int64_t off = 0;
    for (int q = 0; q < r; ++q) off += reinterpret_cast<const int64_t *>(a.msgs + (int64_t)q * a.msg_bytes)[1];
    return off;
}

// rows: one thread per node ENTRY of a rank (coalesced reads of the narrow ids, coalesced 8-byte stores); the first `rows`
// threads of a rank also place edge_ptr = rank offset + rank-local offset.  The rank's offset is summed once per block from
// the message headers.
__global__ __launch_bounds__(256) void ugs_collate_rows(CollateArgs a) {
    __shared__ int64_t off_sh;
    const int r = (int)blockIdx.y;
    if (threadIdx.x == 0) off_sh = edge_offset(a, r);
    __syncthreads();
    const int64_t off = off_sh;
    const unsigned char *m = a.msgs + (int64_t)r * a.msg_bytes;
    const int64_t rows = a.row_off[r + 1] - a.row_off[r];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows * a.k) a.nodes[a.row_off[r] * a.k + i] = widen(m + a.off_nodes, i, a.node_b, true);
    if (i < rows) a.edge_ptr[a.row_off[r] + i] = off + (int64_t)reinterpret_cast<const uint32_t *>(m + a.off_eptr)[i];
    if (r == a.world - 1 && i == 0) a.edge_ptr[a.row_off[a.world]] = off + reinterpret_cast<const int64_t *>(m)[1];
    // a total above edge_cap means the sender truncated its message: the caller compares this word with its capacity, lazily
    if (a.max_total && i == 0) atomicMax(a.max_total, (long long)reinterpret_cast<const int64_t *>(m)[1]);
}

// edge entries: a block takes 1024 consecutive entries of one rank, four per thread at a stride of 256 (coalesced reads of the
// narrow wire types, coalesced 8-byte stores)
__global__ __launch_bounds__(256) void ugs_collate_edges(CollateArgs a) {
    __shared__ int64_t off_sh;
    const int r = (int)blockIdx.y;
    const unsigned char *m = a.msgs + (int64_t)r * a.msg_bytes;
    const int64_t tot = reinterpret_cast<const int64_t *>(m)[1];
    const int64_t e_lo = (int64_t)blockIdx.x * 1024;
    if (e_lo >= tot || e_lo >= a.edge_cap) return;                 // uniform per block
    if (threadIdx.x == 0) off_sh = edge_offset(a, r);
    __syncthreads();
    const int64_t off = off_sh;
    const bool sgn = a.eidx_b != 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t e = e_lo + q * 256 + threadIdx.x;
        if (e >= tot || e >= a.edge_cap) continue;
        const int64_t pos = off + e;
        if (pos >= a.ld) continue;                                 // ld is also the capacity of the output buffers
        a.edge_index[pos] = widen(m + a.off_eidx, e, a.eidx_b, sgn);
        a.edge_index[a.ld + pos] = widen(m + a.off_eidx, a.edge_cap + e, a.eidx_b, sgn);
        a.edge_src[pos] = widen(m + a.off_esrc, e, a.esrc_b, true);
    }
}

}  // namespace

hipError_t ugs_launch_collate_unpack(const void *d_msgs, int world, int64_t msg_bytes, const int64_t *row_off, int k, int node_b, int eidx_b,
                                     int esrc_b, int64_t rows_cap, int64_t edge_cap, const int64_t *section_off4, int64_t *d_nodes,
                                     int64_t *d_edge_index, int64_t ld, int64_t *d_edge_ptr, int64_t *d_edge_src, int64_t *d_max_total,
                                     hipStream_t s) {
    CollateArgs a{};
    a.msgs = static_cast<const unsigned char *>(d_msgs);
    a.msg_bytes = msg_bytes;
    a.world = world; a.k = k; a.node_b = node_b; a.eidx_b = eidx_b; a.esrc_b = esrc_b;
    a.rows_cap = rows_cap; a.edge_cap = edge_cap;
    a.off_nodes = section_off4[0]; a.off_eptr = section_off4[1]; a.off_eidx = section_off4[2]; a.off_esrc = section_off4[3];
    for (int r = 0; r <= world; ++r) a.row_off[r] = row_off[r];
    a.nodes = d_nodes; a.edge_index = d_edge_index; a.edge_ptr = d_edge_ptr; a.edge_src = d_edge_src; a.ld = ld;
    a.max_total = reinterpret_cast<long long *>(d_max_total);
    const unsigned gx_rows = (unsigned)(((rows_cap > 0 ? rows_cap : 1) * (int64_t)k + 255) / 256);
    hipLaunchKernelGGL(ugs_collate_rows, dim3(gx_rows, (unsigned)world), dim3(256), 0, s, a);
    if (edge_cap > 0) {
        const unsigned gx_e = (unsigned)((edge_cap + 1023) / 1024);
        hipLaunchKernelGGL(ugs_collate_edges, dim3(gx_e, (unsigned)world), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

// ---- streamed host-visible calls (ugs_sample_batch_stream): a chunk's edge_ptr starts at 0; the caller's starts at the chunk's base ----
namespace {
__global__ __launch_bounds__(256) void ugs_rebase_edge_ptr(const int64_t *__restrict__ in, int64_t *__restrict__ out, int64_t n, int64_t base) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] + base;
}
}  // namespace

hipError_t ugs_launch_rebase_edge_ptr(const int64_t *in, int64_t *out, int64_t n, int64_t base, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ugs_rebase_edge_ptr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n, base);
    return hipGetLastError();
}
