"""Soak of the streamed calls (ugs_sample_batch_stream / ugs_sample_stream): random batches, row counts, k, modes and chunk sizes, each
call compared tensor by tensor with the two-phase call of the same product (which the parity suites hold to the oracle); shapes alternate
so that pooled device buffers and pinned host blocks are reused at other sizes; some calls start early on a wrong guess.
usage: python tools/streamed_soak.py [iterations] [seed]  -> one JSON line"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ugs_sampler  # noqa: E402
from ugs_sampler._lib import lib  # noqa: E402
import ctypes as C  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
nrng = np.random.default_rng(rng.randrange(1 << 30))
ugs_sampler._STREAM_MIN_ROWS = 1


def small_batch():
    G = rng.randint(1, 12)
    cols, ptr = [], [0]
    for _ in range(G):
        n = rng.choice([4, 7, 12, 20, 33, 60])
        off = ptr[-1]
        p = rng.choice([0.15, 0.3, 0.6])
        e = [(u + off, v + off) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
        cols += e + [(v, u) for u, v in e]
        ptr.append(off + n)
    if not cols:
        cols = [(0, 1), (1, 0)]
    return np.ascontiguousarray(np.array(cols, dtype=np.int64).T), np.array(ptr, dtype=np.int64)


def er(n, deg):
    e = nrng.integers(0, n, size=(2, n * deg // 2), dtype=np.int64)
    return np.ascontiguousarray(e[:, e[0] != e[1]]), np.array([0, n], dtype=np.int64)


def stats():
    a, b = C.c_int64(), C.c_int64()
    lib.ugs_stream_stats(C.byref(a), C.byref(b))
    return a.value, b.value


pool = [small_batch() for _ in range(6)] + [er(2000, 16), er(1200, 60), er(5000, 8)]
bad, calls, handle_calls = [], 0, 0
for it in range(iters):
    ei, ptr = pool[rng.randrange(len(pool))]
    modified = rng.random() < 0.15
    if modified:                                  # a batch that agrees with a remembered one in the sampled words only
        ei = ei.copy()
        j = rng.randrange(ei.shape[1])
        ei[:, j] = ei[:, (j + 1) % ei.shape[1]]
    G = len(ptr) - 1
    k = rng.choice([2, 3, 4, 6, 8])
    m = rng.choice([1, 17, 200, 1500]) if G > 1 else rng.choice([300, 2500, 9000])
    mode = rng.choice(["sample", "graph", "global"])
    seed = rng.randrange(-5, 1000)
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    early = None
    if modified and (ei.shape[1], G, m, k, mode) in ugs_sampler._stream_totals:      # streamed FIRST: its early start picks the unmodified batch's plan
        os.environ["UGS_STREAM_CHUNK_ROWS"] = str(rng.choice([64, 333, 100000]))
        early = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=seed)
    os.environ["UGS_NO_STREAMED_CALL"] = "1"
    want = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=seed)
    del os.environ["UGS_NO_STREAMED_CALL"]
    if early is not None and not all(torch.equal(a, b) for a, b in zip(early, want)):
        bad.append(("batch, streamed before the two-phase call", it, ei.shape[1], G, m, k, mode, seed))
    os.environ["UGS_STREAM_CHUNK_ROWS"] = str(rng.choice([1, 7, 64, 333, 1000, 100000]) if G * m < 3000 else rng.choice([64, 333, 1000, 100000]))
    ugs_sampler._stream_totals[(ei.shape[1], G, m, k, mode)] = want[1].shape[1] if rng.random() < 0.9 else max(0, want[1].shape[1] // 2)
    got = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=seed)
    calls += 1
    if not all(torch.equal(a, b) for a, b in zip(got, want)):
        bad.append(("batch", it, ei.shape[1], G, m, k, mode, seed, os.environ["UGS_STREAM_CHUNK_ROWS"]))
    if G == 1 and rng.random() < 0.5:
        n = int(ptr[-1])
        h = ugs_sampler.create_preproc(ei_t, n, k)
        em, off = rng.choice([("local", 0), ("flat", 0), ("global", rng.randrange(0, 10 ** 6))])
        os.environ["UGS_NO_STREAMED_CALL"] = "1"
        want = ugs_sampler.sample(h, m, k, em, off, seed)
        del os.environ["UGS_NO_STREAMED_CALL"]
        ugs_sampler._stream_totals[("handle", int(h), m, k, em)] = want[1].shape[1]
        got = ugs_sampler.sample(h, m, k, em, off, seed)
        handle_calls += 1
        if not all(torch.equal(a, b) for a, b in zip(got, want)):
            bad.append(("handle", it, n, m, k, em, off, seed, os.environ["UGS_STREAM_CHUNK_ROWS"]))
        ugs_sampler.destroy_preproc(h)
    if it % 50 == 49:
        print(f"iteration {it + 1}: {len(bad)} mismatches", file=sys.stderr, flush=True)
kept, wrong = stats()
print(json.dumps({"iterations": iters, "batch_calls": calls, "handle_calls": handle_calls, "mismatches": len(bad), "first_mismatches": bad[:5],
                  "early_starts_kept": kept, "early_starts_thrown_away": wrong}))
