"""Two host threads making streamed calls at the same time on one device, over the SAME few batches (shared cached plans, shared job and
copy streams): every result compared with the two-phase call's, computed beforehand on one thread.
usage: python tools/streamed_threads.py [calls per thread]  -> one JSON line"""
import json
import os
import random
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ugs_sampler  # noqa: E402

per_thread = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ugs_sampler._STREAM_MIN_ROWS = 1
nrng = np.random.default_rng(3)


def er(n, deg):
    e = nrng.integers(0, n, size=(2, n * deg // 2), dtype=np.int64)
    return np.ascontiguousarray(e[:, e[0] != e[1]]), np.array([0, n], dtype=np.int64)


def small(seed):
    r = random.Random(seed)
    cols, ptr = [], [0]
    for _ in range(8):
        n = r.choice([6, 11, 25, 40])
        off = ptr[-1]
        e = [(u + off, v + off) for u in range(n) for v in range(u + 1, n) if r.random() < 0.4]
        cols += e + [(v, u) for u, v in e]
        ptr.append(off + n)
    return np.ascontiguousarray(np.array(cols, dtype=np.int64).T), np.array(ptr, dtype=np.int64)


cases = []
for (ei, ptr), m, k in ((er(3000, 20), 6000, 8), (er(1500, 70), 3000, 6), (small(1), 400, 4), (small(2), 900, 5)):
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    os.environ["UGS_NO_STREAMED_CALL"] = "1"
    wants = {seed: ugs_sampler.sample_batch(ei_t, ptr_t, m, k, seed=seed) for seed in range(6)}
    del os.environ["UGS_NO_STREAMED_CALL"]
    ugs_sampler._stream_totals[(ei.shape[1], len(ptr) - 1, m, k, "sample")] = max(w[1].shape[1] for w in wants.values())
    cases.append((ei_t, ptr_t, m, k, wants))
os.environ["UGS_STREAM_CHUNK_ROWS"] = "500"
bad, done = [], [0, 0]


def work(tid):
    r = random.Random(100 + tid)
    ugs_sampler._select_device(None, jobs=True)
    for i in range(per_thread):
        ei_t, ptr_t, m, k, wants = cases[r.randrange(len(cases))]
        seed = r.randrange(6)
        got = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, seed=seed)
        if not all(torch.equal(a, b) for a, b in zip(got, wants[seed])):
            bad.append((tid, i, m, k, seed))
        done[tid] += 1


ts = [threading.Thread(target=work, args=(t,)) for t in range(2)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print(json.dumps({"threads": 2, "calls": done, "mismatches": len(bad), "first": bad[:5]}))
