"""Synthetic workloads for the sampler benchmarks and parity tests.

No dataset can be downloaded here, so the BASELINE.json configurations are realised as *TU-shaped*
synthetic batches (same node/edge statistics as MUTAG / PROTEINS / QM9, both edge directions stored
like PyG's ToUndirected, reference src/gps/gps/datasets.py:150) and a seeded Erdos-Renyi graph.
"""
import random

import numpy as np

# name -> (nodes per graph, undirected edges per graph, k, graphs per call, samples per graph)
TU_SHAPES = {
    "c1_mutag_b32": (18, 20, 4, 32, 1),
    "c2_mutag_b1024": (18, 20, 4, 32, 32),
    "c3_proteins_b8192": (39, 73, 6, 32, 256),
    "c4_qm9_b65536": (18, 19, 5, 32, 2048),
}


def tu_graph(n, n_und_edges, seed):
    """One connected TU-shaped graph: random spanning tree over a shuffled vertex order, then random
    extra non-loop, non-duplicate pairs up to `n_und_edges`; returns int64 [2, 2*n_und_edges] holding all
    (u,v) with u<v sorted, then all (v,u)."""
    rng = random.Random(seed)
    order = list(range(n))
    rng.shuffle(order)
    und = set()
    for i in range(1, n):
        a, b = order[i], order[rng.randrange(i)]
        und.add((min(a, b), max(a, b)))
    max_e = n * (n - 1) // 2
    target = min(max(n_und_edges, len(und)), max_e)
    while len(und) < target:
        a, b = rng.randrange(n), rng.randrange(n)
        if a != b:
            und.add((min(a, b), max(a, b)))
    e = sorted(und)
    us = [u for u, _ in e] + [v for _, v in e]
    vs = [v for _, v in e] + [u for u, _ in e]
    return np.array([us, vs], dtype=np.int64).reshape(2, -1)


def tu_batch(n, n_und_edges, num_graphs, dataset_seed=0, first_graph=0):
    """A PyG-style batch of `num_graphs` TU-shaped graphs: (edge_index int64 [2,E], ptr int64 [G+1])."""
    cols, ptr = [], [0]
    for g in range(num_graphs):
        ei = tu_graph(n, n_und_edges, dataset_seed * 1000003 + first_graph + g)
        cols.append(ei + ptr[-1])
        ptr.append(ptr[-1] + n)
    ei = np.concatenate(cols, axis=1) if cols else np.zeros((2, 0), np.int64)
    return np.ascontiguousarray(ei), np.array(ptr, dtype=np.int64)


def er_graph(n, n_cols, seed=0):
    """Erdos-Renyi-style multigraph of SURVEY.md C5: `n_cols` columns with u, v ~ U[0, n) drawn from
    torch.Generator().manual_seed(seed) (u first, then v), self loops dropped, duplicates kept, one column
    per undirected edge (the sampler symmetrises).  Returns (edge_index int64 [2,E'], ptr [0, n])."""
    import torch

    g = torch.Generator().manual_seed(seed)
    u = torch.randint(0, n, (n_cols,), generator=g, dtype=torch.int64)
    v = torch.randint(0, n, (n_cols,), generator=g, dtype=torch.int64)
    keep = u != v
    ei = torch.stack([u[keep], v[keep]]).contiguous().numpy()
    return ei, np.array([0, n], dtype=np.int64)


def workload(name):
    """Returns (edge_index, ptr, m_per_graph, k) for a BASELINE.json configuration name."""
    if name in TU_SHAPES:
        n, e, k, G, m = TU_SHAPES[name]
        ei, ptr = tu_batch(n, e, G)
        return ei, ptr, m, k
    if name == "c5_er_1m":
        ei, ptr = er_graph(1_000_000, 20_000_000, 0)
        return ei, ptr, 1_000_000, 8
    if name.startswith("er_"):   # er_<n>_<cols>[_<m>_<k>]  (proxy sizes for tests)
        parts = name.split("_")
        ei, ptr = er_graph(int(parts[1]), int(parts[2]), 0)
        m, k = (int(parts[3]), int(parts[4])) if len(parts) >= 5 else (int(parts[1]), 8)
        return ei, ptr, m, k
    raise KeyError(name)


def algorithmic_bytes(nodes, edge_ptr, k, csr_degree_of_node):
    """Mean algorithmic HBM bytes per sample, SURVEY.md section 8(d):
    16 [alias prob+idx, order] + sum_{v in S}(16 + 4 deg v) [indptr pair + adjacency once per sampled vertex]
    + sum_{v in S[0:k-1]} 4 deg v [index_of gather per scanned neighbour] + 4 Es [edge_col read]
    + 8k + 8 + 24 Es [nodes, edge_ptr, edge_index(2 x i64) + edge_src(i64) writes].
    `csr_degree_of_node[v]` = CSR degree (after symmetrisation) of output node id v."""
    nodes = np.asarray(nodes)
    valid = nodes >= 0
    deg = np.where(valid, csr_degree_of_node[np.where(valid, nodes, 0)], 0).astype(np.int64)
    es = np.diff(np.asarray(edge_ptr)).astype(np.int64)
    per = 16 + (16 * valid.sum(1) + 4 * deg.sum(1)) + 4 * deg[:, : max(k - 1, 0)].sum(1) + 4 * es + 8 * k + 8 + 24 * es
    return float(per.mean()) if len(per) else 0.0
