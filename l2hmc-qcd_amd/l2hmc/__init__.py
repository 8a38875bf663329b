"""l2hmc -- MI355X-native drop-in for the PyTorch hot path of saforem2/l2hmc-qcd.

Same import names as the reference package (``l2hmc.dynamics.pytorch.dynamics.Dynamics``,
``l2hmc.lattice.su3.pytorch.lattice.LatticeSU3``, ...), but the lattice / group / network
arithmetic runs in hand-written gfx950 HIP kernels (``libl2q.so``, C ABI in ``include/l2q.h``).

Unlike the reference's ``__init__`` (src/l2hmc/__init__.py:11-52) nothing here needs
mpi4py: rank / world size come from the torchrun-style environment.
"""
from __future__ import annotations

import logging
import os

import torch

__version__ = '0.1.0'

RANK = int(os.environ.get('RANK', 0))
LOCAL_RANK = int(os.environ.get('LOCAL_RANK', 0))
WORLD_SIZE = int(os.environ.get('WORLD_SIZE', 1))

# reference: src/l2hmc/__init__.py:46-52 decides DEVICE at import
DEVICE = 'cuda' if torch.cuda.is_available() else 'cpu'


def get_logger(name: str | None = None) -> logging.Logger:
    log = logging.getLogger(name)
    if not logging.getLogger().handlers:
        logging.basicConfig(level=logging.INFO if RANK == 0 else logging.ERROR,
                            format='[%(asctime)s][%(levelname)s][%(name)s] %(message)s')
    return log
