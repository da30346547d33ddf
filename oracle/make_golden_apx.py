#!/usr/bin/env python3
"""oracle/make_golden_apx.py -- TEST INFRASTRUCTURE ONLY.  Generates tests/golden/apx_ugs.json from the REFERENCE
apx_ugs_sampler (oracle/_ref, built unmodified from /root/reference by oracle/build_ref.py:build_apx).  The reference needs
10-30 s per call (10^6-trial rejection loops), so only a few tiny cases are recorded."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

CASES = [
    {"cols": [[0, 1], [1, 2], [2, 0], [2, 3]], "ptr": [0, 4], "m": 1, "k": 3, "seed": 42, "epsilon": 0.5},
    {"cols": [[0, 1], [1, 2], [2, 0], [2, 3], [3, 4], [4, 0]], "ptr": [0, 6], "m": 1, "k": 3, "seed": 7, "epsilon": 0.9},
    {"cols": [[0, 1], [1, 2], [2, 0], [2, 3], [3, 4], [4, 0]], "ptr": [1, 5], "m": 1, "k": 3, "seed": 3, "epsilon": 0.9},   # column sub-range
    {"cols": [[0, 1]], "ptr": [0, 1], "m": 2, "k": 3, "seed": 1, "epsilon": 0.1},                                             # n < k
]

if __name__ == "__main__":
    import torch
    import build_ref
    build_ref.build_apx()
    ref = build_ref.load_apx()
    out = []
    for c in CASES:
        ei = torch.tensor(c["cols"], dtype=torch.long).t().contiguous()
        s, p = ref.sample_batch(ei, torch.tensor(c["ptr"]), c["m"], c["k"], "sample", c["seed"], c["epsilon"])
        out.append(dict(c, samples=s.tolist(), sample_ptr=p.tolist()))
        print(c, "->", s.tolist())
    with open(os.path.join(os.path.dirname(HERE), "tests", "golden", "apx_ugs.json"), "w") as f:
        json.dump(out, f, indent=1)
