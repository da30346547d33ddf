"""Multi-GPU sharding of one sampling job: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards without any data-path collective: result row b = g*m + i depends only on (graph g, seed, i)
(reference src/sampler.cpp:158-176: one RNG per sample, no shared state), so every rank samples a disjoint contiguous
range of the G*m rows against its own HBM-resident copy of the plan.  The only exchange step is the collation of the
finished batch on the rank that feeds the trainer (BASELINE.json north_star: "RCCL gather over xGMI only to collate the
final batch").  xGMI is point-to-point and per-link bound, so the payload is narrowed before it travels: node ids and
edge_src as int32 when they fit, per-row edge counts instead of offsets, local edge ids (mode "sample") as uint8 --
about 3.4x fewer bytes than the int64 tensors -- and everything of a rank goes in ONE message.
"""
import torch
import torch.distributed as dist


def shard_range(total_rows, rank, world):
    """Contiguous row range of `rank`: sizes differ by at most one row."""
    base, rem = divmod(int(total_rows), int(world))
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def _wire_dtypes(k, mode, node_id_bound, edge_id_bound, col_bound):
    small = torch.int32
    nd = small if node_id_bound < 2 ** 31 else torch.int64
    if mode in ("sample", "local", 0):
        ed = torch.uint8 if k <= 255 else small
    else:
        ed = small if edge_id_bound < 2 ** 31 else torch.int64
    sd = small if col_bound < 2 ** 31 else torch.int64
    return nd, ed, sd


def _pack(parts):
    """concatenate tensors of mixed dtypes into one uint8 buffer (each part padded to 16 bytes)."""
    chunks, layout, off = [], [], 0
    for t in parts:
        b = t.contiguous().view(torch.uint8).reshape(-1)
        pad = (-b.numel()) % 16
        if pad:
            b = torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=b.device)])
        layout.append((off, t.numel(), t.dtype, tuple(t.shape)))
        off += b.numel()
        chunks.append(b)
    return torch.cat(chunks) if chunks else torch.zeros(0, dtype=torch.uint8), layout, off


def _unpack(buf, layout):
    out = []
    for off, numel, dtype, shape in layout:
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        out.append(buf[off:off + nbytes].view(dtype).reshape(shape))
    return out


def collate(local, k, mode, node_id_bound, edge_id_bound, col_bound, group=None, dst=0, all_ranks=False):
    """Collates per-rank results (nodes [r,k], edge_index [2,>=t], edge_ptr [r+1], edge_src [>=t]; int64, rank order =
    row order) into the full (nodes, edge_index, edge_ptr, edge_src) on rank `dst` (None elsewhere), or on every rank if
    `all_ranks`.  Works on the tensors' own device (GPU with nccl/RCCL, CPU with gloo)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nodes, edge_index, edge_ptr, edge_src = local
    dev = nodes.device
    rows = nodes.size(0)
    # 1. sizes (one tiny all-gather; the host needs them to size the messages)
    mine = torch.stack([torch.tensor(rows, dtype=torch.int64, device=dev), edge_ptr[-1].to(torch.int64)]).reshape(1, 2)
    sizes = torch.empty((world, 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.cpu().tolist()
    max_rows, max_tot = max(s[0] for s in sizes), max(s[1] for s in sizes)
    tot = sizes[rank][1]
    nd, ed, sd = _wire_dtypes(k, mode, node_id_bound, edge_id_bound, col_bound)
    # 2. narrow + pad to the common message shape
    w_nodes = torch.zeros((max_rows, k), dtype=nd, device=dev)
    w_nodes[:rows] = nodes.to(nd)
    w_counts = torch.zeros((max_rows,), dtype=torch.int32, device=dev)
    w_counts[:rows] = (edge_ptr[1:] - edge_ptr[:-1]).to(torch.int32)
    w_eidx = torch.zeros((2, max_tot), dtype=ed, device=dev)
    w_eidx[:, :tot] = edge_index[:, :tot].to(ed)
    w_esrc = torch.zeros((max_tot,), dtype=sd, device=dev)
    w_esrc[:tot] = edge_src[:tot].to(sd)
    msg, layout, nbytes = _pack([w_nodes, w_counts, w_eidx, w_esrc])
    # 3. the exchange step
    if all_ranks:
        gathered = torch.empty((world, nbytes), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, msg.reshape(1, -1), group=group)
        blocks = [gathered[r] for r in range(world)]
    else:
        blocks = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
        dist.gather(msg, blocks, dst=dst, group=group)
        if rank != dst:
            return None
    # 4. widen + compact in rank (= row) order
    n_l, c_l, e_l, s_l = [], [], [], []
    for r in range(world):
        bn, bc, be, bs = _unpack(blocks[r], layout)
        rr, tt = sizes[r]
        n_l.append(bn[:rr])
        c_l.append(bc[:rr])
        e_l.append(be[:, :tt])
        s_l.append(bs[:tt])
    out_nodes = torch.cat(n_l).to(torch.int64)
    counts = torch.cat(c_l).to(torch.int64)
    out_eptr = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=out_eptr[1:])
    out_eidx = torch.cat(e_l, dim=1).to(torch.int64)
    out_esrc = torch.cat(s_l).to(torch.int64)
    return out_nodes, out_eidx, out_eptr, out_esrc


def default_row_sampler(edge_index, ptr, k):
    """rows -> (nodes, edge_index, edge_ptr, edge_src) on the current GPU through a device-resident Plan (HIP path)."""
    from . import Plan
    plan = Plan.from_batch(edge_index, ptr, k)

    def run(m_per_graph, mode, seed, row_begin, row_count):
        return plan.sample_rows(m_per_graph, mode=mode, seed=seed, row_begin=row_begin, row_count=row_count)
    run.plan = plan
    return run


def sample_batch_sharded(edge_index, ptr, m_per_graph, k, mode="sample", seed=42, group=None, dst=0, all_ranks=True,
                         row_sampler=None):
    """sample_batch with the G*m rows sharded over the ranks of `group`; returns the reference's 5-tuple
    (nodes, edge_index, edge_ptr, sample_ptr, edge_src_global) on `dst` (or on all ranks), identical to the
    single-process result.  `row_sampler(m, mode, seed, row_begin, row_count)` produces a rank's rows; the default is
    the HIP plan path (tests inject a CPU stand-in to exercise the collation over gloo)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    G = int(ptr.numel()) - 1
    rows = max(G, 0) * int(m_per_graph)
    begin, count = shard_range(rows, rank, world)
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
