"""One-off soak of the device batch pass: N random mini-batches (tests/test_gpu_batch_pass.py's generator) through the pass against
the oracle with one LRU history per side -- fused variant, then the two-kernel variant forced by UGS_BP_FUSED_WORK=0."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, p) for p in ("tests", "oracle", "ss-gnn_amd")]
os.environ["UGS_DEVICE_BATCH"] = "1"
import numpy as np, torch, oracle, ugs_sampler
import test_gpu_batch_pass as tb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for tag, env in (("fused", None), ("two-kernel", "0")):
    if env is None: os.environ.pop("UGS_BP_FUSED_WORK", None)
    else: os.environ["UGS_BP_FUSED_WORK"] = env
    rng = random.Random(777 + (env is not None)); pool = []
    ugs_sampler.clear_cache(); cache = oracle.Cache()
    s0 = ugs_sampler.batch_pass_stats()
    for it in range(N):
        ei, ptr, m, k, mode, seed = tb._batch(rng, pool)
        want = oracle.sample_batch(ei, ptr, m, k, mode, seed, cache)
        got = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), m, k, mode, seed)
        for g, w in zip(got, want):
            assert np.array_equal(g.numpy(), np.asarray(w)), (tag, it)
    s1 = ugs_sampler.batch_pass_stats()
    print(tag, N, "batches bit-exact; device plans", s1["device_plans"] - s0["device_plans"], "general path", s1["general_path"] - s0["general_path"], flush=True)
    cache.close()
