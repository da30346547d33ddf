#!/usr/bin/env python3
"""oracle/make_golden.py -- TEST INFRASTRUCTURE ONLY.

Generates tests/golden/*.npz by running every scenario of tests/scenarios.py against the REFERENCE
itself (oracle/_ref, built from /root/reference by oracle/build_ref.py), one fresh process per scenario
(the reference's preprocessing LRU is process-global).  Runs only in the build container; the fixtures
(inputs + expected outputs, data only) are committed so the GPU box needs neither the reference nor this
script.  Also writes tests/golden/f3_ring_uniformity_stats.json, the statistics the reference's
tests/test_uniformity.py prints for the synthetic ring.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)


def run_one(name):
    import scenarios as sc
    from backends import RefBackend
    calls = sc.SCENARIOS[name]()
    results = sc.run_scenario(calls, RefBackend())
    sc.save_golden(name, calls, results)
    if name == "f3_ring_uniformity":
        from uniformity_stats import script_stats, true_stats
        nodes, edge_index, edge_ptr, _, _ = results[0]
        with open(os.path.join(sc.GOLDEN_DIR, "f3_ring_uniformity_stats.json"), "w") as f:
            json.dump({"script": script_stats(nodes, edge_index, edge_ptr, 4),
                       "true": true_stats(nodes, edge_index, edge_ptr, 4)}, f, indent=1)
    print(name, "->", len(calls), "calls")


if __name__ == "__main__":
    import scenarios as sc
    if len(sys.argv) > 1:
        run_one(sys.argv[1])
    else:
        import build_ref
        build_ref.build()
        for name in sc.SCENARIOS:
            subprocess.run([sys.executable, os.path.abspath(__file__), name], check=True)
