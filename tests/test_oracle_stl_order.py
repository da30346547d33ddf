"""CPU: the oracle's restatement of libstdc++ std::unordered_set<int> iteration order against the REAL
container of the host toolchain (oracle/stl_probe.cpp), through the 13 -> ... -> 10273 bucket chain."""
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(HERE), "oracle")


@pytest.fixture(scope="module")
def probe():
    if shutil.which("g++") is None:
        pytest.skip("no g++ to build the libstdc++ probe")
    subprocess.run(["make", "-C", ORACLE_DIR, "stl_probe"], check=True, capture_output=True)
    return os.path.join(ORACLE_DIR, "stl_probe")


def test_bucket_chain(probe):
    out = subprocess.run([probe, "chain"], check=True, capture_output=True, text=True).stdout.split()
    got = [int(x) for x in out[1::2]]
    want = [13, 29, 59, 127, 257, 541, 1109, 2357, 5087, 10273, 20753, 42043, 85229, 172933, 351061, 712697,
            1447153, 2938679, 5967347]
    assert got[:len(want)] == want


def test_iteration_order_matches_real_unordered_set(probe):
    rng = random.Random(123)
    seqs = []
    for n in [0, 1, 2, 5, 12, 13, 14, 15, 28, 29, 30, 31, 59, 60, 127, 128, 129, 257, 258, 541, 542, 600, 1109, 1110,
              2400, 6000, 12000]:
        for rep in range(3):
            hi = rng.choice([max(n, 1) * 2, 1000, 1_000_000, 2**31 - 1])
            s = [rng.randrange(hi) for _ in range(n)]
            if rep == 2 and n > 3:   # heavy duplication + clustered residues
                s = [x - x % 13 for x in s] + s[: n // 2]
            seqs.append(s)
    text = "".join(f"{len(s)} " + " ".join(map(str, s)) + "\n" for s in seqs)
    lines = subprocess.run([probe, "order"], input=text, check=True, capture_output=True, text=True).stdout.strip().split("\n")
    assert len(lines) == len(seqs)
    for s, line in zip(seqs, lines):
        want = [int(x) for x in line.split()][1:]
        got = oracle.stl_order(s).tolist()
        assert got == want, (len(s), got[:10], want[:10])
