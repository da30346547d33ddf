"""Multi-GPU sharding of one sampling job: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards without any data-path collective: result row b = g*m + i depends only on (graph g, seed, i)
(reference src/sampler.cpp:158-176: one RNG per sample, no shared state), so every rank samples a disjoint contiguous
range of the G*m rows against its own HBM-resident copy of the plan.  The only exchange step is the collation of the
finished batch on the rank that feeds the trainer (BASELINE.json north_star: "RCCL gather over xGMI only to collate the
final batch").

`Collator` is that step.  It is set up ONCE per job with fixed capacities, so a steady-state step needs no host round trip:
  * every rank packs its rows into ONE fixed-size message (wire format: include/ugs_mi355.h, ugs_collate_layout).  xGMI is
    point-to-point and per-link bound, so the payload is narrowed before it travels -- node ids and edge_src as int32 when
    they fit, rank-local uint32 edge offsets, local edge ids (mode "sample") as uint8: about 3.4x fewer bytes than the int64
    tensors; the conversion writes straight into the message (no padded staging copies);
  * one gather (or all-gather) of those messages;
  * the destination turns the `world` messages into the batch's int64 tensors in one pass, with the edge offsets of the ranks
    computed ON the device from the message headers: a HIP kernel pair on the GPU (ugs_collate_unpack), torch index
    operations with the same result on the CPU (gloo tests).
Outputs keep their capacity: edge_index [2, cap] / edge_src [cap] are valid up to edge_ptr[-1], which stays on the device.
"""
import ctypes as C

import torch
import torch.distributed as dist


def shard_offsets(total_rows, world, weights=None):
    """First row of every rank's contiguous range, plus the total: world + 1 non-decreasing offsets.

    weights=None: the equal split (sizes differ by at most one row).  Otherwise `weights[r]` is the share of rank r (any
    non-negative numbers, e.g. rows per millisecond measured in a warm-up): rank r gets total * w[r] / sum(w) rows, rounded so
    that the ranges cover [0, total) exactly (largest remainders first, ties to the lower rank) -- a deterministic function
    of its arguments, so every rank computes the same split from the same all-gathered weights.  Any split is legal: row i
    depends only on (graph, seed, i) (reference src/sampler.cpp:158-161)."""
    total, world = int(total_rows), int(world)
    if weights is None:
        base, rem = divmod(total, world)
        sizes = [base + (1 if r < rem else 0) for r in range(world)]
    else:
        w = [float(x) for x in weights]
        if len(w) != world or any(not (x >= 0.0) or x == float("inf") for x in w) or not sum(w) > 0.0:
            raise ValueError("shard weights: one finite non-negative number per rank, not all zero")
        # exact rational arithmetic: every float is a ratio of integers, so the quotas total * w[r] / sum(w) are compared and
        # floored without rounding -- the sizes always add up to `total` (floating-point floors could overshoot it)
        from fractions import Fraction
        fw = [Fraction(x) for x in w]
        tot_w = sum(fw)
        exact = [total * x / tot_w for x in fw]
        sizes = [e.numerator // e.denominator for e in exact]
        left = total - sum(sizes)
        assert 0 <= left < world or (left == 0 and world == 0)
        order = sorted(range(world), key=lambda r: (-(exact[r] - sizes[r]), r))
        for r in order[:left]:
            sizes[r] += 1
    off = [0]
    for c in sizes:
        off.append(off[-1] + c)
    assert off[-1] == total
    return off


def shard_range(total_rows, rank, world, weights=None):
    """Contiguous row range (begin, count) of `rank` under shard_offsets(total_rows, world, weights)."""
    off = shard_offsets(total_rows, world, weights)
    return off[rank], off[rank + 1] - off[rank]


def _wire_dtypes(k, mode, node_id_bound, edge_id_bound, col_bound):
    small = torch.int32
    nd = small if node_id_bound < 2 ** 31 else torch.int64
    if mode in ("sample", "local", 0):
        ed = torch.uint8 if k <= 255 else small
    else:
        ed = small if edge_id_bound < 2 ** 31 else torch.int64
    sd = small if col_bound < 2 ** 31 else torch.int64
    return nd, ed, sd


def _layout(k, nb, eb, sb, rows_cap, edge_cap):
    """(section offsets [nodes, edge_ptr, edge_index, edge_src], message bytes) of the wire format -- asked from the library, so
    that the Python packer and the HIP unpacker cannot disagree."""
    from ._lib import check, lib
    so, mb = (C.c_int64 * 4)(), C.c_int64()
    check(lib.ugs_collate_layout(k, nb, eb, sb, int(rows_cap), int(edge_cap), so, C.byref(mb)))
    return list(so), mb.value


class Collator:
    """Collates the ranks' results of one sharded job, step after step, without host synchronisation.

    total_rows        G * m_per_graph of the job; rank r owns rows [row_off[r], row_off[r+1])
    row_off           world + 1 offsets (shard_offsets(total_rows, world, weights)); None = the equal split.  Uneven splits are
                      how the destination rank -- which also unpacks the whole batch -- is given fewer rows to sample
    edge_cap          capacity in edge entries of ONE rank's result: an UPPER BOUND, the same number on every rank (rows_cap *
                      2k(k-1) always holds; a probe step plus a margin is the practical choice -- then poll overflowed())
    node_id_bound / edge_id_bound / col_bound   exclusive bounds of the values in nodes / edge_index / edge_src (wire widths)
    dst / all_ranks   the batch is produced on rank `dst` only (gather), or on every rank (all-gather)

    collate(local) takes this rank's (nodes [rows,k], edge_index [2,>=t], edge_ptr [rows+1], edge_src [>=t]) int64 tensors and
    returns, on the destination(s), (nodes [B,k], edge_index [2,world*edge_cap], edge_ptr [B+1], edge_src [world*edge_cap]) --
    buffers owned by the collator, overwritten by the next call -- and None elsewhere.

    Overflow: a step whose edge total on some rank exceeds edge_cap cannot be represented (the message is fixed-size); the
    entries beyond the capacity are dropped and the step's result is INVALID.  This is detected on the device without a host
    round trip: `max_total` (a one-word device tensor) holds the largest per-rank total seen -- by pack() on every rank for its
    own totals, by unpack() on the destination for all ranks' -- and overflowed() / check() read it back (one host
    synchronisation: call them lazily, e.g. once per epoch, or before trusting a batch)."""

    def __init__(self, total_rows, k, mode, node_id_bound, edge_id_bound, col_bound, edge_cap, device, group=None, dst=0,
                 all_ranks=False, world=None, rank=None, row_off=None):
        self.group, self.dst, self.all_ranks = group, dst, all_ranks
        if world is None:
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:                                                  # explicit geometry: pack / unpack without a process group (tests)
            self.world, self.rank = int(world), int(rank)
        if self.world > 64:
            raise RuntimeError("Collator: at most 64 ranks")
        self.k, self.total_rows, self.edge_cap = int(k), int(total_rows), int(edge_cap)
        self.dev = torch.device(device)
        self.row_off = shard_offsets(total_rows, self.world) if row_off is None else [int(x) for x in row_off]
        if len(self.row_off) != self.world + 1 or self.row_off[0] != 0 or self.row_off[-1] != self.total_rows or \
                any(self.row_off[r + 1] < self.row_off[r] for r in range(self.world)):
            raise RuntimeError("Collator: row_off must be world + 1 non-decreasing offsets from 0 to total_rows")
        self.rows = self.row_off[self.rank + 1] - self.row_off[self.rank]
        self.rows_cap = max(self.row_off[r + 1] - self.row_off[r] for r in range(self.world))
        self.nd, self.ed, self.sd = _wire_dtypes(k, mode, node_id_bound, edge_id_bound, col_bound)
        self.nb, self.eb, self.sb = (torch.empty((), dtype=d).element_size() for d in (self.nd, self.ed, self.sd))
        self.sec, self.nbytes = _layout(self.k, self.nb, self.eb, self.sb, self.rows_cap, self.edge_cap)
        self.msg = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.dev)
        self._views = self._sections(self.msg)
        self._views[0][0] = self.rows                          # header: this rank's row count never changes
        self.max_total = torch.zeros(1, dtype=torch.int64, device=self.dev)   # largest per-rank edge total seen (device; see overflowed())
        self.is_dst = all_ranks or self.rank == dst
        if self.is_dst:
            self.inbox = torch.zeros((self.world, self.nbytes), dtype=torch.uint8, device=self.dev)
            cap = self.world * self.edge_cap
            self.out_nodes = torch.empty((self.total_rows, self.k), dtype=torch.int64, device=self.dev)
            self.out_eptr = torch.empty((self.total_rows + 1,), dtype=torch.int64, device=self.dev)
            # (the torch path keeps one sink slot behind the capacity: where the entries of an overflowing step are dropped)
            sink = 0 if self.dev.type == "cuda" else 1
            self._eidx_buf = torch.empty((2, max(cap, 1) + sink), dtype=torch.int64, device=self.dev)
            self._esrc_buf = torch.empty((max(cap, 1) + sink,), dtype=torch.int64, device=self.dev)
            self.out_eidx = self._eidx_buf[:, :max(cap, 1)]
            self.out_esrc = self._esrc_buf[:max(cap, 1)]
            self._row_off_c = (C.c_int64 * (self.world + 1))(*self.row_off)
            self._lanes = torch.arange(self.edge_cap, dtype=torch.int64, device=self.dev) if self.dev.type != "cuda" else None

    def _sections(self, buf):
        """typed views (header, nodes, edge_ptr, edge_index, edge_src) of one message"""
        o_n, o_p, o_e, o_s = self.sec
        rc, ec, k = self.rows_cap, self.edge_cap, self.k
        head = buf[0:16].view(torch.int64)
        nodes = buf[o_n:o_n + rc * k * self.nb].view(self.nd).reshape(rc, k)
        eptr = buf[o_p:o_p + (rc + 1) * 4].view(torch.int32)
        eidx = buf[o_e:o_e + 2 * ec * self.eb].view(self.ed).reshape(2, ec)
        esrc = buf[o_s:o_s + ec * self.sb].view(self.sd)
        return head, nodes, eptr, eidx, esrc

    # -- this rank's rows -> its message (dtype-converting copies straight into the message; nothing is synchronised) ----------
    def pack(self, local):
        nodes, edge_index, edge_ptr, edge_src = local
        rows = self.rows
        if nodes.size(0) != rows:
            raise RuntimeError(f"Collator: rank {self.rank} owns {rows} rows, got {nodes.size(0)}")
        head, w_nodes, w_eptr, w_eidx, w_esrc = self._views
        head[1:2].copy_(edge_ptr[rows:rows + 1])
        torch.maximum(self.max_total, edge_ptr[rows:rows + 1], out=self.max_total)
        w_nodes[:rows].copy_(nodes)
        w_eptr[:rows + 1].copy_(edge_ptr)                      # values < 2^32: stored as the low 32 bits
        n = min(self.edge_cap, edge_index.size(1))             # whatever lies beyond the step's total is never read
        if n:
            w_eidx[:, :n].copy_(edge_index[:, :n])
            w_esrc[:n].copy_(edge_src[:n])
        return self.msg

    def exchange(self):
        if self.dev.type == "cuda" and dist.get_backend(self.group) == "gloo":
            return self._exchange_through_host()
        if self.all_ranks:
            dist.all_gather_into_tensor(self.inbox, self.msg.reshape(1, -1), group=self.group)
        else:
            blocks = [self.inbox[r] for r in range(self.world)] if self.is_dst else None
            dist.gather(self.msg, blocks, dst=self.dst, group=self.group)

    def _exchange_through_host(self):
        """device messages over a CPU-only backend (gloo): staged through host memory -- a rehearsal path (several ranks on
        one GPU), not the production path, which hands the device buffers to RCCL"""
        msg = self.msg.cpu()
        if self.all_ranks:
            inbox = torch.empty((self.world, self.nbytes), dtype=torch.uint8)
            dist.all_gather_into_tensor(inbox, msg.reshape(1, -1), group=self.group)
            self.inbox.copy_(inbox)
        else:
            blocks = [torch.empty(self.nbytes, dtype=torch.uint8) for _ in range(self.world)] if self.is_dst else None
            dist.gather(msg, blocks, dst=self.dst, group=self.group)
            if self.is_dst:
                self.inbox.copy_(torch.stack(blocks))

    # -- `world` messages -> the batch ---------------------------------------------------------------------------------------
    def unpack(self):
        if not self.is_dst:
            return None
        if self.dev.type == "cuda":
            from ._lib import check, lib
            check(lib.ugs_collate_unpack(self.inbox.data_ptr(), self.world, self._row_off_c, self.k, self.nb, self.eb, self.sb,
                                         self.rows_cap, self.edge_cap, self.out_nodes.data_ptr(), self.out_eidx.data_ptr(),
                                         self.out_eidx.stride(0), self.out_eptr.data_ptr(), self.out_esrc.data_ptr(),
                                         self.max_total.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream))
        else:
            self._unpack_torch()
        return self.out_nodes, self.out_eidx, self.out_eptr, self.out_esrc

    def _unpack_torch(self):
        """the same result with torch operations (CPU / gloo): the offsets stay tensors, rank blocks are placed in rank order
        with computed indices, a block's slack beyond its total is overwritten by the next rank's entries"""
        off = torch.zeros((), dtype=torch.int64, device=self.dev)
        for r in range(self.world):
            head, w_nodes, w_eptr, w_eidx, w_esrc = self._sections(self.inbox[r])
            r0, r1 = self.row_off[r], self.row_off[r + 1]
            rows = r1 - r0
            self.out_nodes[r0:r1].copy_(w_nodes[:rows])
            lp = w_eptr[:rows].to(torch.int64) & 0xFFFFFFFF
            torch.add(lp, off, out=self.out_eptr[r0:r1])
            if self.edge_cap:
                idx = torch.clamp(self._lanes + off, max=self.out_esrc.numel())      # beyond the capacity (overflow): the sink slot
                blk = w_eidx.to(torch.int64)
                if self.ed == torch.uint8:
                    blk = blk & 0xFF
                self._eidx_buf.index_copy_(1, idx, blk)
                self._esrc_buf.index_copy_(0, idx, w_esrc.to(torch.int64))
            off = off + head[1]
            torch.maximum(self.max_total, head[1:2], out=self.max_total)
        self.out_eptr[self.total_rows:].copy_(off.reshape(1))

    # -- capacity check (lazy: one host synchronisation per call) ---------------------------------------------------------------
    def overflowed(self):
        """True if some step so far had a per-rank edge total above edge_cap (that step's batch was truncated: invalid)."""
        return int(self.max_total.item()) > self.edge_cap

    def check(self):
        t = int(self.max_total.item())
        if t > self.edge_cap:
            raise RuntimeError(f"Collator: a rank produced {t} edge entries in one step, capacity is {self.edge_cap}: that step's batch is invalid")

    def collate(self, local):
        self.pack(local)
        self.exchange()
        return self.unpack()

    __call__ = collate


def collate(local, k, mode, node_id_bound, edge_id_bound, col_bound, group=None, dst=0, all_ranks=False):
    """One-off collation with exact-size results (setup cost and two host round trips per call: use a Collator in a loop).
    Per-rank (nodes [r,k], edge_index [2,>=t], edge_ptr [r+1], edge_src [>=t]) -> the full tensors on `dst` / every rank."""
    world = dist.get_world_size(group)
    nodes, edge_index, edge_ptr, edge_src = local
    dev = nodes.device
    mine = torch.stack([torch.tensor(nodes.size(0), dtype=torch.int64, device=dev), edge_ptr[-1].to(torch.int64)]).reshape(1, 2)
    sizes = torch.empty((world, 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.cpu()
    total_rows, edge_cap = int(sizes[:, 0].sum()), int(sizes[:, 1].max())
    row_off = [0]
    for c_ in sizes[:, 0].tolist():                            # any contiguous split in rank order (equal or weighted)
        row_off.append(row_off[-1] + int(c_))
    c = Collator(total_rows, k, mode, node_id_bound, edge_id_bound, col_bound, edge_cap, dev, group=group, dst=dst, all_ranks=all_ranks,
                 row_off=row_off)
    res = c.collate(local)
    if res is None:
        return None
    out_nodes, out_eidx, out_eptr, out_esrc = res
    tot = int(sizes[:, 1].sum())
    return out_nodes, out_eidx[:, :tot].contiguous(), out_eptr, out_esrc[:tot].contiguous()


def default_row_sampler(edge_index, ptr, k):
    """rows -> (nodes, edge_index, edge_ptr, edge_src) on the current GPU through a device-resident Plan (HIP path)."""
    from . import Plan
    plan = Plan.from_batch(edge_index, ptr, k)

    def run(m_per_graph, mode, seed, row_begin, row_count):
        return plan.sample_rows(m_per_graph, mode=mode, seed=seed, row_begin=row_begin, row_count=row_count)
    run.plan = plan
    return run


def sample_batch_sharded(edge_index, ptr, m_per_graph, k, mode="sample", seed=42, group=None, dst=0, all_ranks=True,
                         row_sampler=None, weights=None):
    """sample_batch with the G*m rows sharded over the ranks of `group`; returns the reference's 5-tuple
    (nodes, edge_index, edge_ptr, sample_ptr, edge_src_global) on `dst` (or on all ranks), identical to the
    single-process result.  `row_sampler(m, mode, seed, row_begin, row_count)` produces a rank's rows; the default is
    the HIP plan path (tests inject a CPU stand-in to exercise the collation over gloo)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    G = int(ptr.numel()) - 1
    rows = max(G, 0) * int(m_per_graph)
    begin, count = shard_range(rows, rank, world, weights)     # weights: the same list on every rank (shard_offsets)
    if row_sampler is None:
        row_sampler = default_row_sampler(edge_index, ptr, k)
    local = row_sampler(int(m_per_graph), mode, int(seed), begin, count)
    node_bound = int(ptr[-1]) if ptr.numel() else 0
    edge_bound = max(node_bound, int(m_per_graph) * int(k))
    res = collate(local, k, mode, node_bound, edge_bound, int(edge_index.size(1)), group=group, dst=dst, all_ranks=all_ranks)
    if res is None:
        return None
    nodes, eidx, eptr, esrc = res
    sample_ptr = torch.arange(G + 1, dtype=torch.int64, device=nodes.device) * int(m_per_graph)
    return nodes, eidx, eptr, sample_ptr, esrc
