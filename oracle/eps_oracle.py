"""oracle/eps_oracle.py -- TEST INFRASTRUCTURE ONLY.

Exact output DISTRIBUTION of the reference's epsilon_uniform_sampler for small graphs, by enumerating its random process
(reference src/samplers/epsilon_uniform_sampler/src/epsilon_uniform_sampler.cpp): start vertex uniform (:33-35), then
repeatedly a uniform frontier vertex (:57-58); no neighbour outside the sample -> the frontier vertex is erased (:68-72),
otherwise a uniform entry of its candidate list (adjacency order, duplicates kept, :61-66,75) is appended to the sample and
the frontier and the tracked weight is multiplied by 1/|frontier| * 1/|candidates| (:80-81); an attempt that reaches k
vertices is accepted with probability min(1, eps/(w+eps)) (:238); up to max(10, int(10/eps)) attempts per sample (:207).
The reference's generators are seeded per OpenMP thread, so it has no reproducible output: parity of any implementation
with it is statistical, and this enumeration is the law both must follow.

Parity status of this oracle: pinned statistically against the reference itself (tests/test_eps_oracle.py, chi-square of
the reference's empirical frequencies against this law).
"""
from collections import defaultdict


def adjacency(edge_cols, n):
    """adjacency lists in the reference's order (:163-176): for every column (u,v): adj[u].append(v); adj[v].append(u)."""
    adj = [[] for _ in range(n)]
    for u, v in edge_cols:
        if 0 <= u < n and 0 <= v < n:
            adj[u].append(v)
            adj[v].append(u)
    return adj


def attempt_law(adj, n, k):
    """{(sorted node tuple, weight): probability} over successful attempts, and the failure probability of one attempt."""
    out = defaultdict(float)
    fail = 0.0
    if n < k:
        return out, 1.0

    def rec(nodes, frontier, weight, prob, tries):
        nonlocal fail
        if len(nodes) == k:
            out[(tuple(sorted(nodes)), weight)] += prob
            return
        if tries >= k * 100 or not frontier:
            fail += prob
            return
        fs = len(frontier)
        for fi, u in enumerate(frontier):
            cands = [v for v in adj[u] if v not in nodes]
            p_f = prob / fs
            if not cands:
                rec(nodes, frontier[:fi] + frontier[fi + 1:], weight, p_f, tries + 1)
                continue
            for v in cands:
                nf = frontier + (v,)
                rec(nodes + (v,), nf, weight * ((1.0 / len(nf)) * (1.0 / len(cands))), p_f / len(cands), tries + 1)

    for s in range(n):
        rec((s,), (s,), 1.0 * (1.0 / n), 1.0 / n, 0)
    return out, fail


def sample_law(adj, n, k, epsilon):
    """{sorted node tuple: probability that a sample row equals it} and the probability of a failed (-1) row."""
    law, _ = attempt_law(adj, n, k)
    per_attempt = defaultdict(float)
    for (nodes, w), p in law.items():
        per_attempt[nodes] += p * min(1.0, epsilon / (w + epsilon))
    p_succ = sum(per_attempt.values())
    attempts = max(10, int(10.0 / epsilon))
    p_fail_row = (1.0 - p_succ) ** attempts
    scale = (1.0 - p_fail_row) / p_succ if p_succ > 0 else 0.0
    return {t: p * scale for t, p in per_attempt.items()}, p_fail_row


def expected_edges(edge_cols, nodes_sorted, mode, node_offset=0):
    """the edge rows of one successful sample (:265-291): batch columns with both endpoints in the sample, in column order."""
    pos = {v: i for i, v in enumerate(nodes_sorted)}
    rows = []
    for e, (u, v) in enumerate(edge_cols):
        if u in pos and v in pos:
            rows.append((pos[u], pos[v], e) if mode == "sample" else (node_offset + u, node_offset + v, e))
    return rows
