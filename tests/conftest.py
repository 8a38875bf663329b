"""pytest configuration: registers the ``gpu`` marker and puts the product package
(``l2hmc-qcd_amd/`` holds the importable ``l2hmc`` package) and the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'l2hmc-qcd_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:      # pragma: no cover
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _restore_torch_state():
    """Tests (and the Trainer, like the reference's) switch torch's default dtype and the library's
    tuning table; put both back so that no test depends on what ran before it."""
    import torch
    old = torch.get_default_dtype()
    yield
    torch.set_default_dtype(old)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    return load
