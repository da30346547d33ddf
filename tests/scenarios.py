"""Scenario definitions + replayer shared by oracle/make_golden.py and the parity tests.

A *scenario* is a sequence of calls against the `ugs_sampler` surface (reference
src/samplers/ugs_sampler/src/extension.cpp:4-13) made in ONE fresh process, because the reference keeps
a process-global LRU of preprocessing handles whose key ignores `k`
(include/cache.hpp:81-109, src/ugs_sampler_batch_extension.cpp:15-38): results depend on call history.

A golden file tests/golden/<name>.npz stores every call's inputs and the reference's outputs (or, for
large outputs, their SHA-256 + shape), so replaying needs nothing but the .npz.
"""
import hashlib
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
import ugs_workloads as wl  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
DIGEST_ABOVE = 200_000   # int64 elements; larger outputs are stored as digests


def _ei(cols):
    return np.array(cols, dtype=np.int64).T.reshape(2, -1)


def _rand_graph(rng, n, p, both):
    e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
    if both:
        e = e + [(v, u) for u, v in e]
    return _ei(e)


# ---------------------------------------------------------------------------------------------------
# scenario builders: each returns a list of call dicts (inputs only)
# ---------------------------------------------------------------------------------------------------
def sc_f1_sample_batch_test():
    """F1: the reference's tests/test_sample_batch.py batch x modes x seeds."""
    ptr = np.array([0, 4, 8], dtype=np.int64)
    ei = np.array([[0, 1, 2, 4, 5, 6, 4], [1, 2, 3, 5, 6, 7, 7]], dtype=np.int64)
    calls = []
    for mode in ("sample", "graph", "global"):
        for seed in (42, 0, -5):
            calls.append(dict(fn="sample_batch", edge_index=ei, ptr=ptr, m=2, k=3, mode=mode, seed=seed))
    return calls


def sc_f2_handle_api():
    """F2: triangle + tail (reference tests/test_debug_sampling.py:7-10) through the handle API."""
    ei = np.array([[0, 1, 1, 2, 2, 0, 2, 3], [1, 0, 2, 1, 0, 2, 3, 2]], dtype=np.int64)
    calls = [dict(fn="create_preproc", edge_index=ei, num_nodes=4, k=3, slot=0)]
    for em in ("local", "flat", "global"):
        for bo in (0, 10):
            calls.append(dict(fn="sample", slot=0, m=10, k=3, edge_mode=em, base_offset=bo, seed=42))
    calls.append(dict(fn="sample", slot=0, m=7, k=3, edge_mode="local", base_offset=0, seed=-17))
    calls.append(dict(fn="destroy_preproc", slot=0))
    calls.append(dict(fn="sample", slot=0, m=1, k=3, edge_mode="local", base_offset=0, seed=1))  # invalid handle
    return calls


def sc_f3_ring_uniformity():
    """F3: 6-ring, k=4, 5000 samples (reference tests/test_uniformity.py:53-75 synthetic fallback)."""
    ei = np.array([[0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0], [1, 0, 2, 1, 3, 2, 4, 3, 5, 4, 0, 5]], dtype=np.int64)
    return [dict(fn="sample_batch", edge_index=ei, ptr=np.array([0, 6], dtype=np.int64), m=5000, k=4, mode="sample", seed=42)]


def sc_f4_diagnostics():
    """F4: reference tests/test_diagnostics.py graphs (triangle in 4 nodes; chain-5 k=4 one-direction cols;
    single edge in 3 nodes k=3 -> relaxation level 1, every row incomplete) + an edgeless graph (level 2)."""
    calls = []
    tri = np.array([[0, 1, 1, 2, 2, 0], [1, 0, 2, 1, 0, 2]], dtype=np.int64)
    chain = np.array([[0, 1, 2, 3], [1, 2, 3, 4]], dtype=np.int64)
    single = np.array([[0], [1]], dtype=np.int64)
    none = np.zeros((2, 0), dtype=np.int64)
    for slot, (ei, n, k) in enumerate([(tri, 4, 3), (chain, 5, 4), (single, 3, 3), (none, 3, 2), (tri, 4, 1)]):
        calls.append(dict(fn="create_preproc", edge_index=ei, num_nodes=n, k=k, slot=slot))
        calls.append(dict(fn="sample", slot=slot, m=12, k=k, edge_mode="local", base_offset=0, seed=42))
        calls.append(dict(fn="sample", slot=slot, m=5, k=k, edge_mode="global", base_offset=100, seed=3))
    calls.append(dict(fn="sample_batch", edge_index=single, ptr=np.array([0, 3], dtype=np.int64), m=6, k=3, mode="sample", seed=42))
    calls.append(dict(fn="sample_batch", edge_index=none, ptr=np.array([0, 3, 5], dtype=np.int64), m=3, k=2, mode="graph", seed=42))
    return calls


def sc_f5_cache_quirks():
    """F5: 10x repeated triangle batch twice (cache-hit path, reference tests/test_cache_performance.py:19-32),
    then the SAME graph with k=3 -> k=2 -> k=3 in one process (the LRU key ignores k)."""
    tri = [(0, 1), (1, 0), (1, 2), (2, 1), (2, 0), (0, 2)]
    cols, ptr = [], [0]
    for g in range(10):
        cols += [(u + 3 * g, v + 3 * g) for u, v in tri]
        ptr.append(3 * (g + 1))
    ei, ptr = _ei(cols), np.array(ptr, dtype=np.int64)
    calls = [dict(fn="sample_batch", edge_index=ei, ptr=ptr, m=5, k=3, mode="sample", seed=42),
             dict(fn="sample_batch", edge_index=ei, ptr=ptr, m=5, k=3, mode="sample", seed=42)]
    rng = random.Random(5)
    g = _rand_graph(rng, 12, 0.3, True)
    p1 = np.array([0, 12], dtype=np.int64)
    for k in (4, 2, 5, 4):
        calls.append(dict(fn="sample_batch", edge_index=g, ptr=p1, m=40, k=k, mode="global", seed=42))
    return calls


def sc_f6_degenerate_in_batch():
    """F6: graphs with n<k, n=0, cross-graph columns, out-of-range columns, self loops, unsorted columns."""
    rng = random.Random(6)
    cols, ptr = [], [0]
    for n in (5, 2, 0, 9, 1, 7):
        off = ptr[-1]
        e = [(u + off, v + off) for u in range(n) for v in range(u + 1, n) if rng.random() < 0.5]
        cols += e + [(v, u) for u, v in e]
        if n >= 5:
            cols.append((off + 1, off + 1))         # self loop
        ptr.append(off + n)
    cols += [(0, ptr[-1] - 1), (3, 6)]             # cross-graph columns (ignored by every graph)
    rng.shuffle(cols)
    ei, ptr = _ei(cols), np.array(ptr, dtype=np.int64)
    calls = []
    for mode in ("sample", "graph", "global"):
        calls.append(dict(fn="sample_batch", edge_index=ei, ptr=ptr, m=9, k=4, mode=mode, seed=7))
    # handle API with out-of-range cols (reference tests/test_sampler.py:9-11 relies on the silent skip)
    ei2 = np.array([[0, 1, 2, 7, -1, 3], [1, 2, 3, 1, 2, 9]], dtype=np.int64)
    calls.append(dict(fn="create_preproc", edge_index=ei2, num_nodes=4, k=3, slot=0))
    calls.append(dict(fn="sample", slot=0, m=9, k=3, edge_mode="flat", base_offset=0, seed=42))
    return calls


def sc_f7_dense_rehash():
    """F7: dense G(150, 0.6), k=8 -> cut sets of 100+ vertices (bucket chain 13 -> 29 -> 59 -> 127 -> 257)."""
    rng = random.Random(77)
    g = _rand_graph(rng, 150, 0.6, False)
    calls = [dict(fn="create_preproc", edge_index=g, num_nodes=150, k=8, slot=0),
             dict(fn="sample", slot=0, m=300, k=8, edge_mode="local", base_offset=0, seed=42),
             dict(fn="sample_batch", edge_index=g, ptr=np.array([0, 150], dtype=np.int64), m=200, k=8, mode="graph", seed=123456789)]
    g2 = _rand_graph(rng, 700, 0.5, False)   # cuts past 541 / 1109 buckets
    calls.append(dict(fn="sample_batch", edge_index=g2, ptr=np.array([0, 700], dtype=np.int64), m=40, k=6, mode="sample", seed=42))
    return calls


def sc_f8_tu_shapes():
    """F8: the BASELINE.json TU-shaped configurations C1..C4 at full size (outputs as digests where large)."""
    calls = []
    for name in ("c1_mutag_b32", "c2_mutag_b1024", "c3_proteins_b8192", "c4_qm9_b65536"):
        ei, ptr, m, k = wl.workload(name)
        calls.append(dict(fn="sample_batch", edge_index=ei, ptr=ptr, m=m, k=k, mode="sample", seed=42, workload=name))
    # uniformity-script shape: one PROTEINS-like graph n in [20,30], k=8, 5000 samples
    # (reference tests/test_ugs_uniformity_proteins.py:32-55)
    ei = wl.tu_graph(26, 48, 4242)
    calls.append(dict(fn="sample_batch", edge_index=ei, ptr=np.array([0, 26], dtype=np.int64), m=5000, k=8, mode="sample", seed=42))
    return calls


def sc_f9_er_proxy():
    """F9: ER proxy of C5: n=20000, 400000 cols (avg CSR degree 40), k=8, m=20000 (digests) + preproc info."""
    ei, ptr = wl.er_graph(20000, 400000, 0)
    # `regen`: the (large) input is not stored; it is regenerated from the seeded generator and its digest checked
    return [dict(fn="create_preproc", edge_index=ei, num_nodes=20000, k=8, slot=0, regen="er_20000_400000"),
            dict(fn="sample_batch", edge_index=ei, ptr=ptr, m=20000, k=8, mode="sample", seed=42, regen="er_20000_400000")]


def sc_f10_qm9_completeness():
    """F10: reference tests/test_ugs_qm9_completeness.py:22-40 shape: 100 graphs x m=10, k=4 then k=5 in one
    process (the k=5 pass silently reuses the k=4 weights)."""
    calls = []
    for k in (4, 5):
        for g in range(100):
            ei = wl.tu_graph(18, 19, 900 + g)
            calls.append(dict(fn="sample_batch", edge_index=ei, ptr=np.array([0, 18], dtype=np.int64), m=10, k=k, mode="sample", seed=42))
    return calls


def sc_f11_random_batches():
    """F11: seeded random multi-graph batches over all modes (general regression net)."""
    rng = random.Random(11)
    calls = []
    for t in range(40):
        G = rng.randint(1, 6)
        cols, ptr = [], [0]
        for _ in range(G):
            n = rng.choice([1, 3, 5, 8, 12, 20, 40])
            p = rng.choice([0.1, 0.2, 0.4, 0.7])
            off = ptr[-1]
            e = [(u + off, v + off) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
            if rng.random() < 0.5:
                e = e + [(v, u) for u, v in e]
            cols += e
            ptr.append(off + n)
        calls.append(dict(fn="sample_batch", edge_index=_ei(cols), ptr=np.array(ptr, dtype=np.int64),
                          m=rng.choice([1, 2, 7, 33]), k=rng.randint(1, 7),
                          mode=rng.choice(["sample", "graph", "global"]), seed=rng.choice([42, 0, -5, 99991, 2147483647, -2147483648])))
    return calls


SCENARIOS = {
    "f1_sample_batch_test": sc_f1_sample_batch_test,
    "f2_handle_api": sc_f2_handle_api,
    "f3_ring_uniformity": sc_f3_ring_uniformity,
    "f4_diagnostics": sc_f4_diagnostics,
    "f5_cache_quirks": sc_f5_cache_quirks,
    "f6_degenerate_in_batch": sc_f6_degenerate_in_batch,
    "f7_dense_rehash": sc_f7_dense_rehash,
    "f8_tu_shapes": sc_f8_tu_shapes,
    "f9_er_proxy": sc_f9_er_proxy,
    "f10_qm9_completeness": sc_f10_qm9_completeness,
    "f11_random_batches": sc_f11_random_batches,
}
# scenarios whose replay is slow on the CPU oracle are still only seconds; all are used on both sides.


# ---------------------------------------------------------------------------------------------------
# running a scenario against a backend and (de)serialising
# ---------------------------------------------------------------------------------------------------
def digest(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int64))
    return hashlib.sha256(a.tobytes()).hexdigest()


class Backend:
    """Adapter: the surface every backend (reference / oracle / product) is driven through."""

    def fresh(self):
        """reset process-global state (LRU cache) -- a fresh process for the reference."""

    def create_preproc(self, edge_index, num_nodes, k):
        raise NotImplementedError

    def info(self, h):
        raise NotImplementedError

    def sample(self, h, m, k, edge_mode, base_offset, seed):
        raise NotImplementedError

    def destroy_preproc(self, h):
        raise NotImplementedError

    def sample_batch(self, edge_index, ptr, m, k, mode, seed):
        raise NotImplementedError


def run_scenario(calls, backend):
    """Returns a list of results, one per call: tuple of np.int64 arrays, a dict (create_preproc info),
    None (destroy), or ("error", message)."""
    backend.fresh()
    slots, out = {}, []
    for c in calls:
        fn = c["fn"]
        try:
            if fn == "sample_batch":
                r = backend.sample_batch(c["edge_index"], c["ptr"], c["m"], c["k"], c["mode"], c["seed"])
                out.append(tuple(np.asarray(x, dtype=np.int64) for x in r))
            elif fn == "create_preproc":
                h = backend.create_preproc(c["edge_index"], c["num_nodes"], c["k"])
                slots[c["slot"]] = h
                out.append(dict(backend.info(h)))
            elif fn == "sample":
                r = backend.sample(slots[c["slot"]], c["m"], c["k"], c["edge_mode"], c["base_offset"], c["seed"])
                out.append(tuple(np.asarray(x, dtype=np.int64) for x in r))
            elif fn == "destroy_preproc":
                backend.destroy_preproc(slots[c["slot"]])
                out.append(None)
            else:
                raise KeyError(fn)
        except RuntimeError as e:   # noqa: PERF203
            out.append(("error", str(e)))
    return out


def save_golden(name, calls, results):
    arrays, meta = {}, []
    for i, (c, r) in enumerate(zip(calls, results)):
        m = {}
        for key, val in c.items():
            if isinstance(val, np.ndarray) and key == "edge_index" and "regen" in c:
                m[key] = "@regen:" + digest(val)
            elif isinstance(val, np.ndarray):
                arrays[f"c{i}_{key}"] = val
                m[key] = "@array"
            else:
                m[key] = val
        if isinstance(r, tuple) and r and isinstance(r[0], str):
            m["result"] = {"kind": "error", "message": r[1]}
        elif isinstance(r, tuple):
            items = []
            for j, a in enumerate(r):
                if a.size > DIGEST_ABOVE:
                    items.append({"shape": list(a.shape), "sha256": digest(a)})
                else:
                    arrays[f"c{i}_out{j}"] = a
                    items.append({"shape": list(a.shape), "sha256": digest(a), "stored": True})
            m["result"] = {"kind": "arrays", "items": items}
        elif isinstance(r, dict):
            m["result"] = {"kind": "info", "info": r}
        else:
            m["result"] = {"kind": "none"}
        meta.append(m)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **arrays)


def load_golden(name):
    """Returns (calls, expected) where expected[i] is a dict {"kind": ..., ...} with arrays materialised."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    calls, expected = [], []
    for i, m in enumerate(meta):
        c = {}
        for key, val in m.items():
            if key == "result":
                continue
            if isinstance(val, str) and val.startswith("@regen:"):
                ei = wl.workload(m["regen"])[0]
                assert digest(ei) == val[len("@regen:"):], f"regenerated input {m['regen']} differs from the fixture's"
                c[key] = ei
            else:
                c[key] = z[f"c{i}_{key}"] if val == "@array" else val
        res = m["result"]
        if res["kind"] == "arrays":
            for j, it in enumerate(res["items"]):
                if it.get("stored"):
                    it["array"] = z[f"c{i}_out{j}"]
        calls.append(c)
        expected.append(res)
    return calls, expected


def check_against_golden(calls, expected, results, what):
    """Bit-exact comparison (integer outputs; Z compared as an exact double)."""
    assert len(results) == len(expected)
    for i, (c, exp, got) in enumerate(zip(calls, expected, results)):
        tag = f"{what}: call {i} {c['fn']}"
        if exp["kind"] == "error":
            assert isinstance(got, tuple) and got and isinstance(got[0], str), f"{tag}: expected error {exp['message']!r}, got {type(got)}"
            assert exp["message"].split("\n")[0] in got[1] or got[1] in exp["message"], f"{tag}: error text {got[1]!r} vs {exp['message']!r}"
        elif exp["kind"] == "arrays":
            assert isinstance(got, tuple) and not (got and isinstance(got[0], str)), f"{tag}: got {got!r}"
            assert len(got) == len(exp["items"]), tag
            for j, (it, a) in enumerate(zip(exp["items"], got)):
                assert list(a.shape) == it["shape"], f"{tag}: output {j} shape {a.shape} vs {it['shape']}"
                if "array" in it:
                    assert np.array_equal(a, it["array"]), f"{tag}: output {j} differs"
                assert digest(a) == it["sha256"], f"{tag}: output {j} digest differs"
        elif exp["kind"] == "info":
            for key, val in exp["info"].items():
                assert got[key] == val, f"{tag}: info[{key}] {got[key]!r} vs {val!r}"
        else:
            assert got is None, tag
