"""CPU (no GPU needed): the C-ABI library loads and exports every symbol include/ugs_mi355.h declares; the host-side
logic of the product (graph preprocessing, handle registry, error paths that precede any GPU work) matches the oracle
bit for bit.  No sampling call is made here -- sampling exists only on the GPU."""
import ctypes
import os
import random
import re

import numpy as np
import pytest
import torch

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ugs_mi355.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ugs_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = ctypes.CDLL(os.path.join(ROOT, "ss-gnn_amd", "csrc", "libugs_mi355.so"))
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in include/ugs_mi355.h but not exported: {missing}"
    import ugs_sampler
    assert set(ugs_sampler._lib.EXPORTS) == declared, "the ctypes binding and the header disagree"


def test_no_torch_types_in_the_abi():
    hdr = open(os.path.join(ROOT, "include", "ugs_mi355.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", hdr, flags=re.S) and "at::" not in hdr


def test_host_preprocessing_matches_oracle():
    import ugs_sampler
    rng = random.Random(3)
    for _ in range(150):
        n = rng.choice([0, 1, 3, 5, 8, 12, 20, 40, 80, 150])
        p = rng.choice([0.05, 0.1, 0.3, 0.6])
        k = rng.randint(1, 8)
        e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
        if rng.random() < 0.5:
            e = e + [(v, u) for u, v in e]
        if n and rng.random() < 0.3:
            e += [(rng.randrange(n),) * 2, (n + 3, 0), (-1, 0)]      # self loop + out-of-range columns (silently skipped)
        ei = np.array(e, dtype=np.int64).T.reshape(2, -1)
        h = ugs_sampler.create_preproc(torch.from_numpy(ei), n, k)
        P = oracle.Preproc(ei, n, k)
        a, b = ugs_sampler.preproc_dump(h), P.dump()
        for key in b:
            assert np.array_equal(a[key], b[key]), (key, n, k)     # incl. exact alias-table doubles
        i1, i2 = ugs_sampler.get_preproc_info(h), P.info()
        assert all(i1[x] == i2[x] for x in i1) and ugs_sampler.has_graphlets(h) == i2["has_graphlets"]
        ugs_sampler.destroy_preproc(h)
        assert ugs_sampler.get_preproc_info(h) == {} and ugs_sampler.has_graphlets(h) is False
        P.close()


def test_non_contiguous_edge_index_and_handles_are_monotone():
    import ugs_sampler
    big = torch.arange(40, dtype=torch.long).reshape(4, 10) % 7
    view = big[1:3]                       # row stride 10, contiguous columns
    tr = big.t()[:, :2].t()               # column stride != 1 -> copied by the shim
    h1 = ugs_sampler.create_preproc(view, 7, 3)
    h2 = ugs_sampler.create_preproc(view.contiguous(), 7, 3)
    h3 = ugs_sampler.create_preproc(tr, 7, 3)
    assert h1 < h2 < h3
    d1, d2 = ugs_sampler.preproc_dump(h1), ugs_sampler.preproc_dump(h2)
    assert all(np.array_equal(d1[k], d2[k]) for k in d1)
    for h in (h1, h2, h3):
        ugs_sampler.destroy_preproc(h)


def test_argument_errors_raised_before_any_gpu_work():
    import ugs_sampler
    ei = torch.tensor([[0, 1], [1, 2]], dtype=torch.long)
    ptr = torch.tensor([0, 3], dtype=torch.long)
    with pytest.raises(RuntimeError, match="mode must be one of: 'sample', 'graph', 'global'"):
        ugs_sampler.sample_batch(ei, ptr, 1, 2, mode="nope")
    with pytest.raises(RuntimeError, match="edge_index must be int64"):
        ugs_sampler.sample_batch(ei.to(torch.int32), ptr, 1, 2)
    with pytest.raises(RuntimeError, match="ptr must be int64"):
        ugs_sampler.sample_batch(ei, ptr.to(torch.float32), 1, 2)
    with pytest.raises(RuntimeError, match="Invalid preproc handle"):
        ugs_sampler.sample(123456789, 1, 2)
    with pytest.raises(TypeError):
        ugs_sampler.sample_batch(ei, ptr, 1, 2, seed=2 ** 31)


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a machine without a GPU")
def test_sampling_fails_loudly_without_a_gpu():
    import ugs_sampler
    ei = torch.tensor([[0, 1], [1, 2]], dtype=torch.long)
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        ugs_sampler.sample_batch(ei, torch.tensor([0, 3]), 1, 2)
    h = ugs_sampler.create_preproc(ei, 3, 2)
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        ugs_sampler.sample(h, 1, 2)


def test_product_does_not_reference_the_oracle():
    """the product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "ss-gnn_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "ugs_oracle" not in txt and "import oracle" not in txt and "oracle/" not in txt, os.path.join(d, f)
