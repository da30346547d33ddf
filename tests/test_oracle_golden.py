"""CPU: the oracle (oracle/ugs_oracle.c) against the committed golden fixtures generated from the reference
(oracle/make_golden.py).  Bit-exact for every integer output; Z is an exact double."""
import pytest

import scenarios as sc
from backends import OracleBackend

NAMES = sorted(sc.SCENARIOS)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    calls, expected = sc.load_golden(name)
    got = sc.run_scenario(calls, OracleBackend())
    sc.check_against_golden(calls, expected, got, f"oracle vs golden {name}")


@pytest.mark.parametrize("name", NAMES)
def test_golden_inputs_match_scenario_definitions(name):
    """the fixture's stored inputs are the ones tests/scenarios.py defines (guards against stale fixtures)."""
    import numpy as np
    calls, _ = sc.load_golden(name)
    fresh = sc.SCENARIOS[name]()
    assert len(calls) == len(fresh)
    for a, b in zip(calls, fresh):
        assert set(a) == set(b)
        for key in a:
            if isinstance(b[key], np.ndarray):
                assert np.array_equal(a[key], b[key])
            else:
                assert a[key] == b[key]


def test_ring_uniformity_known_answer():
    """The statistics reference tests/test_uniformity.py prints for its synthetic 6-ring (k=4, 5000 samples,
    seed 42): 7 'unique', counts 1492/1040/517/510/485/479/477, CV 0.517, POOR."""
    import json
    import os
    from uniformity_stats import script_stats
    calls, _ = sc.load_golden("f3_ring_uniformity")
    (nodes, edge_index, edge_ptr, _, _), = sc.run_scenario(calls, OracleBackend())
    st = script_stats(nodes, edge_index, edge_ptr, 4)
    with open(os.path.join(sc.GOLDEN_DIR, "f3_ring_uniformity_stats.json")) as f:
        gold = json.load(f)["script"]
    assert st == gold
    assert st["counts"] == [1492, 1040, 517, 510, 485, 479, 477] and st["cv"] == 0.517 and st["verdict"] == "POOR"
