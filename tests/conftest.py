import os
import sys
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'))

    def t(self, key):
        return torch.from_numpy(self.z[key])

    def lst(self, key):
        out, i = [], 0
        while '%s.%d' % (key, i) in self.z:
            out.append(torch.from_numpy(self.z['%s.%d' % (key, i)]))
            i += 1
        return out


@pytest.fixture
def golden():
    return Golden


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
