"""GPU: the GPU variant of apx_ugs_sampler.sample_batch (csrc/ugs_apx_gpu.hip) -- statistical parity.

The reference draws everything from one sequential generator, so the GPU variant (one generator per (sample, trial)) cannot be
bit-equal; what must agree is the output LAW.  oracle/apx_oracle.py enumerates that law exactly for k = 3 on small graphs
(tests/test_apx_law.py pins the enumeration against the sequential restatement, which is bit-exact with the reference); here the
GPU rows are tested against it (Pearson chi-square, p > 1e-4), together with everything that is deterministic: the APX-DD order and
bucket estimates the GPU variant used, determinism in (graph, seed), the return shape, connectivity and support of every row.
k = 4 is held to the law of oracle/apx_oracle.py:law_k (exact root and growth factors, acceptance factor by 2*10^5 joint draws of its
36 binomials), k = 9 (beyond the 8 of round 2; 720-permutation cap active) to connectivity and determinism."""
import collections

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

GRAPHS = {"house": [(0, 1), (1, 2), (2, 3), (3, 0), (0, 4), (1, 4)], "kite": [(0, 1), (0, 2), (1, 2), (1, 3), (2, 3), (3, 4)]}


@pytest.mark.parametrize("name", sorted(GRAPHS))
def test_gpu_rows_follow_the_enumerated_law(name):
    import torch
    import apx_oracle as ao
    import apx_ugs_sampler
    edges = GRAPHS[name]
    n = max(max(e) for e in edges) + 1
    adj = ao.adjacency(n, edges)
    eps, k, m = 0.9, 3, 1500
    ei = torch.tensor(edges, dtype=torch.long).t().contiguous()
    ptr = torch.tensor([0, len(edges)])
    s, p, pos, est = apx_ugs_sampler.sample_batch(ei, ptr, m, k, seed=7, epsilon=eps, backend="gpu", return_order=True)
    # the order and the estimates are deterministic on these graphs
    w_pos, w_est = ao.order(adj, k, eps)
    assert pos.tolist() == w_pos and est.tolist() == w_est
    S = s.size(1)
    assert s.shape == (k, S) and s.dtype == torch.int64 and p.tolist() == list(range(S + 1)) and 0 < S <= m
    law, acc = ao.law_k3(adj, w_pos, w_est, eps)
    # failed samples are dropped; their share follows from the per-trial acceptance and the 10^6-trial cap
    p_fail = (1.0 - acc) ** 1_000_000
    assert abs((m - S) - m * p_fail) <= 6.0 * np.sqrt(m * p_fail * (1 - p_fail)) + 3
    rows = [tuple(r) for r in s.t().tolist()]
    for r in rows:          # k distinct vertices, grown along edges: every vertex after the first is adjacent to an earlier one
        assert len(set(r)) == k and all(any(r[j] in adj[r[i]] for i in range(j)) for j in range(1, k))
    counts = collections.Counter(rows)
    pval, chi2, dof = ao.chi_square_p(counts, law)
    assert pval > 1e-4, f"{name}: chi2 {chi2:.1f} on {dof} dof, p = {pval:.2e}; counts {dict(counts)}"
    # deterministic in (graph, seed); another seed gives other rows
    s2, _ = apx_ugs_sampler.sample_batch(ei, ptr, 64, k, seed=7, epsilon=eps, backend="gpu")
    s3, _ = apx_ugs_sampler.sample_batch(ei, ptr, 64, k, seed=7, epsilon=eps, backend="gpu")
    s4, _ = apx_ugs_sampler.sample_batch(ei, ptr, 64, k, seed=8, epsilon=eps, backend="gpu")
    assert torch.equal(s2, s3) and not torch.equal(s2[:, :20], s4[:, :20])


def test_gpu_variant_surface_and_limits():
    import torch
    import apx_ugs_sampler
    ei = torch.tensor(GRAPHS["house"], dtype=torch.long).t().contiguous()
    ptr = torch.tensor([0, 6])
    s, p = apx_ugs_sampler.sample_batch(ei, ptr, 0, 3, backend="gpu")
    assert s.shape == (3, 0) and p.tolist() == [0]
    s, p = apx_ugs_sampler.sample_batch(ei, ptr, 5, 6, backend="gpu")            # fewer vertices than k: nothing to sample
    assert s.shape == (6, 0)
    with pytest.raises(RuntimeError):
        apx_ugs_sampler.sample_batch(ei, ptr, 5, 33, backend="gpu")              # k > 32 is outside the product's limits (UGS_KMAX)
    with pytest.raises(RuntimeError):
        apx_ugs_sampler.sample_batch(ei, ptr, 5, 3, backend="tpu")
    s, p = apx_ugs_sampler.sample_batch(ei, ptr, 40, 4, seed=3, epsilon=0.9, backend="gpu")      # k = 4: rows are connected 4-sets
    adj = {v: set() for v in range(5)}
    for u, v in GRAPHS["house"]:
        adj[u].add(v); adj[v].add(u)
    for r in s.t().tolist():
        assert len(set(r)) == 4 and all(any(r[j] in adj[r[i]] for i in range(j)) for j in range(1, 4))


def test_gpu_rows_follow_the_law_for_k4():
    """k = 4 on the kite: 16 ordered graphlets in the support; the sequential restatement's rows are held to the same law in
    tests/test_apx_law.py (committed counts, generator oracle/make_golden_apx_counts.py)"""
    import torch
    import apx_oracle as ao
    import apx_ugs_sampler
    edges = GRAPHS["kite"]
    adj = ao.adjacency(5, edges)
    eps, k, m = 0.9, 4, 1500
    ei = torch.tensor(edges, dtype=torch.long).t().contiguous()
    s, p, pos, est = apx_ugs_sampler.sample_batch(ei, torch.tensor([0, len(edges)]), m, k, seed=5, epsilon=eps, backend="gpu", return_order=True)
    w_pos, w_est = ao.order(adj, k, eps)
    assert pos.tolist() == w_pos and est.tolist() == w_est
    law, acc = ao.law_k(adj, w_pos, w_est, eps, k)
    S = s.size(1)
    p_fail = (1.0 - acc) ** 1_000_000
    assert abs((m - S) - m * p_fail) <= 6.0 * np.sqrt(m * p_fail * (1 - p_fail)) + 3
    counts = collections.Counter(tuple(r) for r in s.t().tolist())
    pval, chi2, dof = ao.chi_square_p(counts, law)
    assert pval > 1e-4, f"k=4: chi2 {chi2:.1f} on {dof} dof, p = {pval:.2e}; counts {dict(counts)}"


def test_gpu_variant_beyond_k8(monkeypatch):
    """k = 9 on a 10-vertex ring with a chord (8! orders of the non-root vertices: the reference's cap of 720 permutations applies,
    :370): every returned row is a connected 9-set grown along edges, and the call is deterministic.  A complete trial costs 720 x 8
    cut estimates of up to 9 x 100 draws, so the trial cap is lowered from 10^6 to 30 000 for this test (UGS_APX_TRIAL_CAP)."""
    monkeypatch.setenv("UGS_APX_TRIAL_CAP", "30000")
    import torch
    import apx_ugs_sampler
    edges = [(i, (i + 1) % 10) for i in range(10)] + [(0, 5)]
    adj = {v: set() for v in range(10)}
    for u, v in edges:
        adj[u].add(v); adj[v].add(u)
    ei = torch.tensor(edges, dtype=torch.long).t().contiguous()
    ptr = torch.tensor([0, len(edges)])
    s, p = apx_ugs_sampler.sample_batch(ei, ptr, 2, 9, seed=2, epsilon=0.9, backend="gpu")
    s2, _ = apx_ugs_sampler.sample_batch(ei, ptr, 2, 9, seed=2, epsilon=0.9, backend="gpu")
    assert torch.equal(s, s2) and s.shape[0] == 9 and p.tolist() == list(range(s.size(1) + 1))
    for r in s.t().tolist():
        assert len(set(r)) == 9 and all(any(r[j] in adj[r[i]] for i in range(j)) for j in range(1, 9))
