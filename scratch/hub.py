import sys, os, time, random
sys.path.insert(0,'ss-gnn_amd'); sys.path.insert(0,'oracle')
import numpy as np, torch, ugs_sampler, oracle
rng = random.Random(5); n = 6500
e = [(0, v) for v in range(1, 6001)] + [(rng.randrange(1, n), rng.randrange(1, n)) for _ in range(9000)]
e = [(u, v) for u, v in e if u != v]
ei = np.array(e, dtype=np.int64).T.reshape(2, -1)
ptr=np.array([0,n],dtype=np.int64)
for m in (1, 4, 24):
    t=time.time(); got=ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), m, 4, mode="sample", seed=42); dt=time.time()-t
    want=oracle.sample_batch(ei,ptr,m,4,"sample",42)
    print("m",m,"time",round(dt,3),"ok",all(np.array_equal(a.numpy(),b) for a,b in zip(got,want)), flush=True)
