// ugs_preproc.hip -- device-side preprocessing of one large graph.
//
// The O(nnz) parts of the reference's Preproc constructor run here; the O(n), order-sensitive floating-point parts
// (degree order, Z, alias table) stay on the host (ugs_host.cpp) so the result is bit-identical to the host path:
//   * A4 CSR of the symmetrised multigraph, entries in column order
//     (reference src/samplers/ugs_sampler/src/preproc.cpp:17-42): one key per endpoint (row vertex; out-of-range columns
//     get the sentinel key n and sort to the tail), a STABLE radix sort of (key, 2*column + side) -- stability is what
//     keeps every row in (column, source-row-first) order -- degrees by atomics, row pointer by an exclusive scan;
//   * A7 suffix degree of every order position (preproc.cpp:100-112);
//   * A8 "can the root reach k vertices inside its suffix graph" (preproc.cpp:114-170): one lane per root, breadth-first
//     with the <= k reached vertices in a private list (the reference's mark array is membership in that list).
// The radix sort and the scan are hipCUB device primitives (plain library passes, not part of the sampling hot path).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>

#include "ugs_device.h"

namespace {

__global__ void __launch_bounds__(256) pre_keys(const int64_t *__restrict__ src, const int64_t *__restrict__ dst, int64_t E, int64_t n,
                                                uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t *__restrict__ deg) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= E) return;
    const int64_t u = src[j], v = dst[j];
    const bool ok = (uint64_t)u < (uint64_t)n && (uint64_t)v < (uint64_t)n;      // silently skipped columns (preproc.cpp:22)
    const uint32_t ku = ok ? (uint32_t)u : (uint32_t)n, kv = ok ? (uint32_t)v : (uint32_t)n;
    reinterpret_cast<uint2 *>(keys)[j] = make_uint2(ku, kv);
    reinterpret_cast<uint2 *>(vals)[j] = make_uint2((uint32_t)(2 * j), (uint32_t)(2 * j + 1));
    atomicAdd(&deg[ku], 1u);
    atomicAdd(&deg[kv], 1u);
}

__global__ void __launch_bounds__(256) pre_entries(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                   const uint32_t *__restrict__ sorted_vals, int64_t nnz,
                                                   int32_t *__restrict__ nbr, int32_t *__restrict__ col) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= nnz) return;
    const uint32_t t = sorted_vals[a];
    const int64_t j = t >> 1;
    nbr[a] = (int32_t)((t & 1u) ? src[j] : dst[j]);       // the row of entry 2j is src[j]: its neighbour is dst[j]
    col[a] = (int32_t)j;
}

__global__ void __launch_bounds__(256) pre_widen(const uint32_t *__restrict__ in, int64_t *__restrict__ out, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = (int64_t)in[i];
}

// one lane per order position: suffix degree + bounded breadth-first reachability (K = compile-time list bound)
__global__ void __launch_bounds__(256) pre_roots(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ nbr,
                                                 const int32_t *__restrict__ order, const int32_t *__restrict__ rank, int32_t n, int32_t k,
                                                 int32_t *__restrict__ sdeg, uint8_t *__restrict__ reach) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int32_t vi = (int32_t)t;
    const int32_t v = order[vi];
    int32_t c = 0;
    for (uint32_t p = rowptr[v], e = rowptr[v + 1]; p < e; ++p) c += rank[nbr[p]] >= vi;
    sdeg[vi] = c;
    int32_t got[UGS_KMAX];
    int32_t cnt = 1;
    got[0] = v;
    for (int32_t h = 0; h < cnt && cnt < k; ++h) {
        const int32_t u = got[h];
        for (uint32_t p = rowptr[u], e = rowptr[u + 1]; p < e && cnt < k; ++p) {
            const int32_t w = nbr[p];
            if (rank[w] < vi) continue;
            bool seen = false;
            for (int32_t i = 0; i < cnt; ++i) seen |= got[i] == w;
            if (!seen) got[cnt++] = w;
        }
    }
    reach[vi] = cnt >= k ? 1 : 0;
}

// plan adjacency straight from the device CSR: adj = (neighbour, order position of neighbour), adjf = (neighbour, batch column)
__global__ void __launch_bounds__(256) pre_assemble(const int32_t *__restrict__ nbr, const int32_t *__restrict__ col, const int32_t *__restrict__ rank,
                                                    const int64_t *__restrict__ colmap, int64_t nnz, int2 *__restrict__ adj, int2 *__restrict__ adjf) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= nnz) return;
    const int32_t w = nbr[a], c = col[a];
    adj[a] = make_int2(w, rank[w]);
    adjf[a] = make_int2(w, colmap ? (int32_t)colmap[c] : c);
}

inline unsigned blocks_for(int64_t items) { return (unsigned)((items + 255) / 256); }

}  // namespace

struct UgsDevPre {
    int64_t n = 0, E = 0, nnz = 0;
    int64_t *src = nullptr, *dst = nullptr;
    uint32_t *keys = nullptr, *keys2 = nullptr, *vals = nullptr, *vals2 = nullptr, *deg = nullptr, *rowptr = nullptr;
    int32_t *nbr = nullptr, *col = nullptr, *order = nullptr, *rank = nullptr, *sdeg = nullptr;
    uint8_t *reach = nullptr;
    void *tmp = nullptr;
    hipStream_t stream = nullptr;
};

size_t ugs_devpre_bytes(int64_t n, int64_t E) {
    return (size_t)E * 16 + (size_t)E * 2 * 4 * 4 + (size_t)E * 2 * 4 * 2 + (size_t)(n + 2) * 4 * 5 + (size_t)n + ((size_t)64 << 20);
}

void ugs_devpre_free(UgsDevPre *d) {
    if (!d) return;
    void *ps[] = {d->src, d->dst, d->keys, d->keys2, d->vals, d->vals2, d->deg, d->rowptr, d->nbr, d->col, d->order, d->rank, d->sdeg, d->reach, d->tmp};
    for (void *p : ps) if (p) (void)hipFree(p);
    delete d;
}

#define PRE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

// Stage 1: upload the columns, build the CSR on the device, hand the row pointer (n+1 int64) back to the host.
hipError_t ugs_devpre_csr(UgsDevPre **out, const int64_t *h_src, const int64_t *h_dst, int64_t E, int64_t n, hipStream_t s,
                          int64_t *h_rowptr, int64_t *nnz_out) {
    auto *d = new UgsDevPre();
    *out = d;
    d->n = n; d->E = E; d->stream = s;
    const int64_t E1 = E > 0 ? E : 1, M = 2 * E1;
    PRE_TRY(hipMalloc(&d->src, (size_t)E1 * 8));
    PRE_TRY(hipMalloc(&d->dst, (size_t)E1 * 8));
    PRE_TRY(hipMalloc(&d->keys, (size_t)M * 4));
    PRE_TRY(hipMalloc(&d->keys2, (size_t)M * 4));
    PRE_TRY(hipMalloc(&d->vals, (size_t)M * 4));
    PRE_TRY(hipMalloc(&d->vals2, (size_t)M * 4));
    PRE_TRY(hipMalloc(&d->deg, (size_t)(n + 2) * 4));
    PRE_TRY(hipMalloc(&d->rowptr, (size_t)(n + 2) * 4));
    PRE_TRY(hipMemcpyAsync(d->src, h_src, (size_t)E * 8, hipMemcpyHostToDevice, s));
    PRE_TRY(hipMemcpyAsync(d->dst, h_dst, (size_t)E * 8, hipMemcpyHostToDevice, s));
    PRE_TRY(hipMemsetAsync(d->deg, 0, (size_t)(n + 2) * 4, s));
    if (E > 0) pre_keys<<<blocks_for(E), 256, 0, s>>>(d->src, d->dst, E, n, d->keys, d->vals, d->deg);
    PRE_TRY(hipGetLastError());
    int bits = 1;
    while (bits < 32 && ((uint64_t)1 << bits) <= (uint64_t)n) ++bits;          // keys are in [0, n]
    size_t tb_sort = 0, tb_scan = 0;
    PRE_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, d->keys, d->keys2, d->vals, d->vals2, (int)(2 * E), 0, bits, s));
    PRE_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb_scan, d->deg, d->rowptr, (int)(n + 2), s));
    PRE_TRY(hipMalloc(&d->tmp, (tb_sort > tb_scan ? tb_sort : tb_scan) + 256));
    // deg[n] counts the sentinel endpoints, so rowptr[n] = number of kept entries
    PRE_TRY(hipcub::DeviceScan::ExclusiveSum(d->tmp, tb_scan, d->deg, d->rowptr, (int)(n + 2), s));
    if (E > 0) PRE_TRY(hipcub::DeviceRadixSort::SortPairs(d->tmp, tb_sort, d->keys, d->keys2, d->vals, d->vals2, (int)(2 * E), 0, bits, s));
    uint32_t nnz32 = 0;
    PRE_TRY(hipMemcpyAsync(&nnz32, d->rowptr + n, 4, hipMemcpyDeviceToHost, s));
    // the host keeps the row pointer as int64 (n+1 entries)
    int64_t *wide = nullptr;
    PRE_TRY(hipMalloc(&wide, (size_t)(n + 1) * 8));
    pre_widen<<<blocks_for(n + 1), 256, 0, s>>>(d->rowptr, wide, n + 1);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_rowptr, wide, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(wide);
    PRE_TRY(e);
    d->nnz = nnz32;
    *nnz_out = d->nnz;
    (void)hipFree(d->keys); d->keys = nullptr;
    (void)hipFree(d->keys2); d->keys2 = nullptr;
    (void)hipFree(d->vals); d->vals = nullptr;
    const int64_t nz1 = d->nnz > 0 ? d->nnz : 1;
    PRE_TRY(hipMalloc(&d->nbr, (size_t)nz1 * 4));
    PRE_TRY(hipMalloc(&d->col, (size_t)nz1 * 4));
    if (d->nnz > 0) pre_entries<<<blocks_for(d->nnz), 256, 0, s>>>(d->src, d->dst, d->vals2, d->nnz, d->nbr, d->col);
    PRE_TRY(hipGetLastError());
    return hipSuccess;
}

// Stage 2: with the host's degree order, compute suffix degrees and the k-reachability flag of every order position.
hipError_t ugs_devpre_roots(UgsDevPre *d, const int32_t *h_order, const int32_t *h_rank, int k, int32_t *h_sdeg, uint8_t *h_reach) {
    const int64_t n = d->n, n1 = n > 0 ? n : 1;
    hipStream_t s = d->stream;
    PRE_TRY(hipMalloc(&d->order, (size_t)n1 * 4));
    PRE_TRY(hipMalloc(&d->rank, (size_t)n1 * 4));
    PRE_TRY(hipMalloc(&d->sdeg, (size_t)n1 * 4));
    PRE_TRY(hipMalloc(&d->reach, (size_t)n1));
    PRE_TRY(hipMemcpyAsync(d->order, h_order, (size_t)n * 4, hipMemcpyHostToDevice, s));
    PRE_TRY(hipMemcpyAsync(d->rank, h_rank, (size_t)n * 4, hipMemcpyHostToDevice, s));
    if (n > 0) pre_roots<<<blocks_for(n), 256, 0, s>>>(d->rowptr, d->nbr, d->order, d->rank, (int32_t)n, k, d->sdeg, d->reach);
    PRE_TRY(hipGetLastError());
    PRE_TRY(hipMemcpyAsync(h_sdeg, d->sdeg, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    PRE_TRY(hipMemcpyAsync(h_reach, d->reach, (size_t)n, hipMemcpyDeviceToHost, s));
    return hipStreamSynchronize(s);
}

// Stage 3: the host copy of the CSR entries (kept for the handle API, re-assembly into other batches, preproc_dump).
hipError_t ugs_devpre_download(UgsDevPre *d, int32_t *h_nbr, int32_t *h_col) {
    if (d->nnz == 0) return hipSuccess;
    PRE_TRY(hipMemcpyAsync(h_nbr, d->nbr, (size_t)d->nnz * 4, hipMemcpyDeviceToHost, d->stream));
    PRE_TRY(hipMemcpyAsync(h_col, d->col, (size_t)d->nnz * 4, hipMemcpyDeviceToHost, d->stream));
    return hipStreamSynchronize(d->stream);
}

// After stage 3 only (nbr, col, rank) are worth keeping: the first plan assembled from this graph reads them on the device.
void ugs_devpre_trim(UgsDevPre *d) {
    void **ps[] = {(void **)&d->src, (void **)&d->dst, (void **)&d->keys, (void **)&d->keys2, (void **)&d->vals, (void **)&d->vals2, (void **)&d->deg,
                   (void **)&d->rowptr, (void **)&d->order, (void **)&d->sdeg, (void **)&d->reach, &d->tmp};
    for (void **p : ps) if (*p) { (void)hipFree(*p); *p = nullptr; }
}

size_t ugs_devpre_resident_bytes(const UgsDevPre *d) { return (size_t)d->nnz * 8 + (size_t)d->n * 4; }

// adj / adjf of a plan (device pointers to this graph's nnz entries).  `h_colmap` (host, `cols` int64 entries, or null for the
// identity) maps the graph's own column numbers to batch columns.
hipError_t ugs_devpre_assemble(UgsDevPre *d, const int64_t *h_colmap, int64_t cols, int2 *adj, int2 *adjf, hipStream_t s) {
    if (d->nnz == 0) return hipSuccess;
    int64_t *cm = nullptr;
    if (h_colmap && cols > 0) {
        PRE_TRY(hipMalloc(&cm, (size_t)cols * 8));
        hipError_t e = hipMemcpyAsync(cm, h_colmap, (size_t)cols * 8, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { (void)hipFree(cm); return e; }
    }
    pre_assemble<<<blocks_for(d->nnz), 256, 0, s>>>(d->nbr, d->col, d->rank, cm, d->nnz, adj, adjf);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (cm) (void)hipFree(cm);
    return e;
}
