"""Statistics over sampled k-subgraphs, as the reference's print-only uniformity scripts compute them.

`script_stats` restates what reference tests/test_uniformity.py:20-36,77-161 counts and prints -- including
its quirk that, in mode="sample", the LOCAL edge ids (0..k-1) are looked up in a map keyed by the sample's
GLOBAL node ids, so an edge only survives when both local ids happen to be node ids of the sample.  The
reference prints exactly these numbers; a bit-exact sampler must reproduce them.
`true_stats` counts distinct (node set, induced edge set) pairs properly.
Thresholds: CV < 0.15 GOOD, < 0.30 MODERATE, else POOR (test_uniformity.py:152-161,
test_ugs_uniformity_proteins.py:120-126)."""
from collections import Counter

import numpy as np


def _finish(counts, total):
    vals = sorted(counts.values(), reverse=True)
    n_valid = sum(vals)
    mean = n_valid / len(vals) if vals else 0.0
    std = (sum((f - mean) ** 2 for f in vals) / len(vals)) ** 0.5 if vals else 0.0
    cv = std / mean if mean > 0 else float("inf")
    verdict = "GOOD" if cv < 0.15 else ("MODERATE" if cv < 0.30 else "POOR")
    return {"valid": int(n_valid), "incomplete": int(total - n_valid), "unique": len(vals),
            "counts": [int(v) for v in vals], "cv": round(float(cv), 3), "verdict": verdict}


def script_stats(nodes, edge_index, edge_ptr, k):
    nodes, edge_index, edge_ptr = np.asarray(nodes), np.asarray(edge_index), np.asarray(edge_ptr)
    counts = Counter()
    for i in range(nodes.shape[0]):
        row = nodes[i][nodes[i] >= 0]
        if len(row) < k:
            continue
        srt = tuple(sorted(int(x) for x in row))
        pos = {v: j for j, v in enumerate(srt)}
        seg = edge_index[:, int(edge_ptr[i]):int(edge_ptr[i + 1])]
        kept = {tuple(sorted((pos[int(u)], pos[int(v)]))) for u, v in seg.T if int(u) in pos and int(v) in pos}
        counts[(srt, tuple(sorted(kept)))] += 1
    return _finish(counts, nodes.shape[0])


def true_stats(nodes, edge_index, edge_ptr, k):
    nodes, edge_index, edge_ptr = np.asarray(nodes), np.asarray(edge_index), np.asarray(edge_ptr)
    counts = Counter()
    for i in range(nodes.shape[0]):
        row = nodes[i]
        if (row < 0).any() or len(row) < k:
            continue
        seg = row[edge_index[:, int(edge_ptr[i]):int(edge_ptr[i + 1])]]     # local ids -> node ids
        edges = tuple(sorted({tuple(sorted((int(u), int(v)))) for u, v in seg.T}))
        counts[(tuple(sorted(int(x) for x in row)), edges)] += 1
    return _finish(counts, nodes.shape[0])
