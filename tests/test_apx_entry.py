"""CPU: the apx_ugs_sampler entry point (host computation by the reference's construction) against the
fixture generated from the reference itself (oracle/make_golden_apx.py).  Bit-exact: same generator, same draw sequence."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_apx_entry_point_reproduces_reference_fixture():
    import apx_ugs_sampler
    with open(os.path.join(ROOT, "tests", "golden", "apx_ugs.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 3
    for c in cases:
        ei = torch.tensor(c["cols"], dtype=torch.long).t().contiguous()
        s, p = apx_ugs_sampler.sample_batch(ei, torch.tensor(c["ptr"]), c["m"], c["k"], mode="sample", seed=c["seed"], epsilon=c["epsilon"])
        assert s.dtype == torch.int64 and s.shape[0] == c["k"] and p.tolist() == c["sample_ptr"]
        assert s.tolist() == c["samples"], c
