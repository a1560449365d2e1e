import json
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    g = json.load(open(os.path.join(GOLDEN, "golden.json")))
    g["mm"] = dict(np.load(os.path.join(GOLDEN, "mm_fixtures.npz")))
    g["sha"] = dict(np.load(os.path.join(GOLDEN, "sha_fixtures.npz")))
    g["aes_kat"] = np.load(os.path.join(GOLDEN, "aes_kat.npz"))["kat"]
    g["chsha"] = dict(np.load(os.path.join(GOLDEN, "chsha_fixtures.npz")))
    return g


def gen_mm(n, seed=0):
    """The reference generator's algorithm (tests/mm_common/mm_generator.py:42-51): pure Python, so the GPU
    box can rebuild the side-256 inputs without the reference checkout."""
    random.seed(seed)
    m1 = [[random.randint(0, 2**32 - 1) for _ in range(n)] for _ in range(n)]
    m2 = [[random.randint(0, 2**32 - 1) for _ in range(n)] for _ in range(n)]
    return np.array(m1, dtype=np.uint32), np.array(m2, dtype=np.uint32)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as o

    o.build()
    return o
