"""One-off soak of the walk tiers added in round 3 (W: 704 candidates, two-pass final; V: 1408 candidates, three-pass final): random
multigraphs whose walks end with hundreds to thousands of candidates, every call forced into one tier (rows that outgrow it are handed
on by the library), all five tensors against the oracle.  usage: tools/tier_soak.py [calls per tier] [tiers, e.g. 1,2,4]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, p) for p in ("tests", "oracle", "ss-gnn_amd")]
import numpy as np, torch, oracle, ugs_sampler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for tier in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("2", "4", "3", "5")):
    os.environ["UGS_FORCE_TIER"] = tier
    rng = random.Random(4000 + int(tier))
    handed = 0
    for it in range(N):
        nv = rng.choice([1500, 4000, 12000])
        deg = rng.choice([40, 80, 120, 160, 240, 400])
        g = np.random.default_rng(rng.randrange(1 << 30))
        ei = g.integers(0, nv, size=(2, nv * deg // 2), dtype=np.int64)
        if rng.random() < 0.25:
            ei = np.concatenate([ei, ei[::-1]], axis=1)
        ptr = np.array([0, nv], dtype=np.int64)
        m, k = rng.choice([64, 300]), rng.choice([4, 6, 8, 8, 10, 12])
        mode, seed = rng.choice(["sample", "graph", "global"]), rng.choice([42, 0, 987654321])
        ugs_sampler.clear_cache()
        want = oracle.sample_batch(ei, ptr, m, k, mode, seed)
        got = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), m, k, mode, seed)
        for a, b in zip(got, want):
            assert np.array_equal(a.numpy(), np.asarray(b)), (tier, it, nv, deg, m, k, mode, seed)
    print("tier", tier, N, "calls bit-exact", flush=True)
