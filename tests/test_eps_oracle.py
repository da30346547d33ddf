"""CPU, build container only: the exact-enumeration law of oracle/eps_oracle.py against the REFERENCE epsilon_uniform_sampler
itself (oracle/_ref, built from /root/reference by oracle/build_ref.py).  The reference has no reproducible output (per-thread
generators), so the check is a chi-square goodness-of-fit of its empirical frequencies against the enumerated law."""
import os

import numpy as np
import pytest

import eps_oracle

REF_SRC = "/root/reference/src/samplers/epsilon_uniform_sampler"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference sources not present on this machine")

GRAPHS = {
    # name: (n, columns)
    "house": (5, [(0, 1), (1, 2), (2, 3), (3, 0), (0, 4), (1, 4)]),
    "tailed_triangle_both_dirs": (5, [(0, 1), (1, 0), (1, 2), (2, 1), (2, 0), (0, 2), (2, 3), (3, 2), (3, 4), (4, 3)]),
    "star_plus": (6, [(0, 1), (0, 2), (0, 3), (0, 4), (4, 5), (1, 2)]),
}


def chi_square(counts, probs, total):
    """returns (statistic, dof) merging cells with expectation < 5"""
    exp = np.array(probs) * total
    obs = np.array(counts, dtype=np.float64)
    order = np.argsort(exp)
    exp, obs = exp[order], obs[order]
    e_m, o_m, ce, co = [], [], 0.0, 0.0
    for e, o in zip(exp, obs):
        ce += e; co += o
        if ce >= 5:
            e_m.append(ce); o_m.append(co); ce = co = 0.0
    if ce > 0 and e_m:
        e_m[-1] += ce; o_m[-1] += co
    e_m, o_m = np.array(e_m), np.array(o_m)
    return float(((o_m - e_m) ** 2 / e_m).sum()), max(len(e_m) - 1, 1)


def check_rows_against_law(nodes, law, p_fail, what):
    from scipy import stats
    total = nodes.shape[0]
    keys = sorted(law)
    counts = {t: 0 for t in keys}
    fails = 0
    for row in nodes:
        if row[0] < 0:
            fails += 1
            continue
        t = tuple(int(x) for x in row)
        assert t in counts, f"{what}: produced {t}, which the law gives probability 0"
        counts[t] += 1
    stat, dof = chi_square([counts[t] for t in keys] + [fails], [law[t] for t in keys] + [p_fail], total)
    p = 1.0 - stats.chi2.cdf(stat, dof)
    assert p > 1e-4, f"{what}: chi2={stat:.1f} dof={dof} p={p:.2e}"
    return p


@pytest.fixture(scope="module")
def ref():
    import build_ref
    build_ref.build_eps()
    os.environ["OMP_NUM_THREADS"] = "1"
    return build_ref.load_eps()


@pytest.mark.parametrize("name", sorted(GRAPHS))
@pytest.mark.parametrize("k,eps", [(3, 0.1), (4, 0.5), (3, 0.01)])
def test_reference_follows_the_enumerated_law(ref, name, k, eps):
    import torch
    n, cols = GRAPHS[name]
    law, p_fail = eps_oracle.sample_law(eps_oracle.adjacency(cols, n), n, k, eps)
    assert abs(sum(law.values()) + p_fail - 1.0) < 1e-9
    ei = torch.tensor(cols, dtype=torch.long).t().contiguous()
    m = 40000
    nodes, eidx, eptr, sptr, esrc = ref.sample_batch(ei, torch.tensor([0, n]), m, k, "sample", 1234, eps)
    check_rows_against_law(nodes.numpy(), law, p_fail, f"reference {name} k={k} eps={eps}")
    # output format of the first successful row against expected_edges
    nodes, eidx, eptr, esrc = nodes.numpy(), eidx.numpy(), eptr.numpy(), esrc.numpy()
    for r in range(50):
        if nodes[r, 0] < 0:
            assert eptr[r + 1] == eptr[r]
            continue
        want = eps_oracle.expected_edges(cols, [int(x) for x in nodes[r]], "sample")
        got = list(zip(eidx[0, eptr[r]:eptr[r + 1]].tolist(), eidx[1, eptr[r]:eptr[r + 1]].tolist(), esrc[eptr[r]:eptr[r + 1]].tolist()))
        assert got == want
