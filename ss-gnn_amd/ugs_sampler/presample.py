"""Device-resident presample cache: the reference trainer's `--presample` path without its Python loops.

The reference samples every dataset graph once at start-up (gps/experiment.py:379-440: `sampler(edge_index, [0, n], m, k,
mode="sample", seed=cfg.seed + running_index)`, results cloned into a dict) and, for every mini-batch, stitches the cached
results of the batch's graphs together on the host (gps/experiment.py:936-993): nodes re-based by ptr[g], edge ids left local,
edge_src re-based by the number of batch columns that belong to earlier graphs, edge_ptr / sample_ptr accumulated -- Python loops
with `.item()` per graph and per sample.  Here the cache lives in HBM as flat tensors and a batch is assembled by a handful of
device gathers; the sizes that decide the output shapes are known on the host from build time, so nothing synchronises.

    cache = PresampleCache(m, k, device="cuda:0")
    for i, data in enumerate(dataset):                       # start-up, like _setup_presampling
        cache.add(i, data.edge_index, data.num_nodes, seed=cfg.seed + i)
    cache.finalize()
    nodes, edge_index, edge_ptr, sample_ptr, edge_src = cache.load(batch.graph_idx, batch.ptr, batch.edge_index)

`load` returns exactly what the reference's `_load_from_presample_cache` stores on the batch -- including its treatment of
graphs whose presampling failed (m rows of -1 to which ptr[g] is ADDED like to every other row, no edges).
"""
import torch

from . import sample_batch


class PresampleCache:
    def __init__(self, m, k, device):
        self.m, self.k = int(m), int(k)
        self.dev = torch.device(device)
        self._slot = {}                 # graph index -> slot (-1: presampling failed)
        self._parts = []                # per slot: (nodes [m,k], edge_index [2,E], edge_ptr [m+1], edge_src [E]) on the device
        self._final = None

    def add(self, index, edge_index, num_nodes, seed):
        """Presample one graph (reference: experiment.py:403-430); a sampler error marks the graph as failed, like the reference."""
        ptr = torch.tensor([0, int(num_nodes)], dtype=torch.long)
        try:
            nodes, eidx, eptr, _, esrc = sample_batch(edge_index.cpu(), ptr, self.m, self.k, mode="sample", seed=int(seed), device=self.dev)
        except Exception:   # noqa: BLE001  (the reference swallows every exception here)
            self._slot[int(index)] = -1
            return False
        self._slot[int(index)] = len(self._parts)
        self._parts.append((nodes, eidx, eptr, esrc))
        self._final = None
        return True

    def finalize(self):
        m, k, dev = self.m, self.k, self.dev
        S = len(self._parts)
        i64 = dict(dtype=torch.int64, device=dev)
        # slot S is the placeholder of failed graphs: m rows of -1, no edges
        self.nodes = torch.cat([p[0] for p in self._parts] + [torch.full((m, k), -1, **i64)], dim=0)                     # [(S+1)*m, k]
        self.eptr_local = torch.stack([p[2] for p in self._parts] + [torch.zeros(m + 1, **i64)], dim=0)                 # [S+1, m+1]
        self.n_edges_host = [int(p[1].size(1)) for p in self._parts] + [0]                                              # host, no sync later
        self.eidx = torch.cat([p[1] for p in self._parts] + [torch.empty((2, 0), **i64)], dim=1)                         # [2, Etot]
        self.esrc = torch.cat([p[3] for p in self._parts] + [torch.empty((0,), **i64)], dim=0)
        base = [0]
        for n in self.n_edges_host[:-1]:
            base.append(base[-1] + n)
        self.edge_base_host = base                                                                                       # first cached edge of every slot
        self.edge_base = torch.tensor(base, **i64)
        self.n_edges = torch.tensor(self.n_edges_host, **i64)
        self._parts = None
        self._final = True

    def load(self, graph_indices, ptr, batch_edge_index):
        """(nodes_sampled [G*m,k], edge_index_sampled [2,Es], edge_ptr [G*m+1], sample_ptr [G+1], edge_src_global [Es]) of a batch made
        of the cached graphs `graph_indices` (a sequence or tensor of dataset indices, in batch order), as device tensors."""
        if self._final is None:
            self.finalize()
        m, k, dev = self.m, self.k, self.dev
        gi = graph_indices.cpu().flatten().tolist() if torch.is_tensor(graph_indices) else list(graph_indices)
        G = len(gi)
        fail = len(self.n_edges_host) - 1
        slots_h = [self._slot.get(int(i), -1) for i in gi]
        slots_h = [fail if s < 0 else s for s in slots_h]
        total = sum(self.n_edges_host[s] for s in slots_h)                        # known on the host: output shapes need no round trip
        i64 = dict(dtype=torch.int64, device=dev)
        slots = torch.tensor(slots_h, **i64)
        ptr_d = ptr.to(dev, dtype=torch.int64)
        # nodes: the slot's m rows + ptr[g] (the reference adds the offset to every entry, -1 padding included: experiment.py:966)
        rows = (slots * m).repeat_interleave(m) + torch.arange(m, **i64).repeat(G)
        nodes = self.nodes.index_select(0, rows) + ptr_d[:G].repeat_interleave(m).unsqueeze(1)
        # edge_ptr: every graph's local offsets shifted by the edge entries of the graphs before it
        cnt = self.n_edges.index_select(0, slots)
        shift = torch.cumsum(cnt, 0) - cnt
        edge_ptr = torch.empty(G * m + 1, **i64)
        edge_ptr[0] = 0
        edge_ptr[1:] = (self.eptr_local.index_select(0, slots)[:, 1:] + shift.unsqueeze(1)).reshape(-1)
        sample_ptr = torch.arange(G + 1, **i64) * m
        # edges: segment gather of the cached entries; ids stay local (the model adds the sample offsets itself, experiment.py:968-971)
        seg = torch.repeat_interleave(torch.arange(G, **i64), cnt, output_size=total)
        src_pos = self.edge_base.index_select(0, slots).index_select(0, seg) + (torch.arange(total, **i64) - shift.index_select(0, seg))
        edge_index = self.eidx.index_select(1, src_pos)
        # edge_src: + number of batch columns whose source vertex belongs to an earlier graph (experiment.py:937-942, 973-974)
        src = batch_edge_index[0].to(dev)
        owner = torch.bucketize(src, ptr_d[1:], right=True)
        inside = ((owner < G) & (src >= ptr_d[0])).to(torch.float64)
        per_graph = torch.bincount(owner.clamp(max=max(G - 1, 0)), weights=inside, minlength=max(G, 1)).to(torch.int64)
        orig_off = torch.cumsum(per_graph, 0) - per_graph
        edge_src = self.esrc.index_select(0, src_pos) + orig_off.index_select(0, seg)
        return nodes, edge_index, edge_ptr, sample_ptr, edge_src
