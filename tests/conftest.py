import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class Golden:
    """npz fixture with 'prefix/key' names -> dict views."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    def __getitem__(self, k):
        return self.z[k]

    def keys(self):
        return self.z.files

    def sub(self, prefix):
        return {k[len(prefix):]: self.z[k] for k in self.z.files if k.startswith(prefix)}


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


def rel_err(got, ref):
    """max |got-ref| / max(|ref|) -- the 'relative fp32' parity metric (tolerance 1e-4)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    denom = max(np.abs(ref).max(), 1e-30) if ref.size else 1.0
    return (np.abs(got - ref).max() / denom) if ref.size else 0.0
