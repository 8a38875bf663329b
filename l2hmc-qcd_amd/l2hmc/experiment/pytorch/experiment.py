"""``Experiment`` -- API shape of src/l2hmc/experiment/pytorch/experiment.py:141-450
(``Experiment(cfg).evaluate(job_type)``, ``.trainer``) without the tracking back-ends."""
from __future__ import annotations

from typing import Optional

import torch

import l2hmc.configs as cfgs
from l2hmc.trainers.pytorch.trainer import Trainer
from l2hmc.utils import dist as D
from l2hmc.utils.dist import setup_torch


class Experiment:
    def __init__(self, cfg: dict | cfgs.ExperimentConfig, build_networks: bool = True,
                 keep=None, skip=None) -> None:
        self.cfg = cfg
        self.config = cfgs.instantiate(cfg) if isinstance(cfg, dict) else cfg
        # Data parallelism: the MODEL must be identical on every rank, the CHAINS
        # must differ.  (1) everything is seeded with the base seed, the Trainer builds networks
        # and numpy masks from it; (2) rank 0's parameters / buffers and masks are broadcast (DDP
        # constructor semantics, trainers/pytorch/trainer.py:246-257 of the reference) so that a
        # nondeterministic initialiser cannot split the replicas; (3) only then torch / cuda /
        # numpy are reseeded per rank (`chain_seed`), so lattice.random(), momenta and accept
        # uniforms are independent streams.  The reference seeds everything with
        # seed * (rank + 1) (utils/dist.py:340), which also makes its numpy masks rank-dependent.
        self._rank = setup_torch(seed=self.config.seed, backend=self.config.backend,
                                 port=self.config.port)
        self.trainer = Trainer(self.config, build_networks=build_networks)
        self.lattice = self.trainer.lattice
        D.sync_model(self.trainer.dynamics)
        D.seed_everything(D.chain_seed(self.config.seed, self._rank))

    def build_trainer(self, **kw) -> Trainer:
        return self.trainer

    def set_net_weights(self, net_weights: cfgs.NetWeights) -> None:
        """Scale the (s, t, q) outputs of every leapfrog network (experiment.py:227-238)."""
        dyn = self.trainer.dynamics
        for step in range(self.config.dynamics.nleapfrog):
            dyn._get_xnet(step, first=True).set_net_weight(net_weights.x)
            dyn._get_xnet(step, first=False).set_net_weight(net_weights.x)
            dyn._get_vnet(step).set_net_weight(net_weights.v)

    def visualize_model(self, x: Optional[torch.Tensor] = None):
        """The reference renders the autograd graph of one xnet / vnet call with torchviz
        (experiment.py:240-271); the networks here run as HIP kernels without an autograd graph,
        so there is nothing to draw."""
        raise NotImplementedError('visualize_model: no autograd graph on the HIP kernel path '
                                  '(and torchviz is not a dependency)')

    def train(self, x: Optional[torch.Tensor] = None, nera: Optional[int] = None,
              nepoch: Optional[int] = None, beta: Optional[float] = None,
              nsteps: Optional[int] = None) -> dict:
        """experiment.py:380-417: the trainer's era / epoch loop with the annealed beta."""
        return self.trainer.train(x=x, beta=beta, nsteps=nsteps, nera=nera, nepoch=nepoch)

    def evaluate(self, job_type: str, beta: Optional[float] = None, nsteps: Optional[int] = None,
                 eps: Optional[float] = None, nleapfrog: Optional[int] = None,
                 x: Optional[torch.Tensor] = None) -> Optional[dict]:
        """'eval' (trained sampler) or 'hmc' (baseline); rank 0 only like the reference
        (experiment.py:419-420)."""
        if self._rank != 0:
            return None
        return self.trainer.eval(beta=beta, x=x, job_type=job_type, nsteps=nsteps, eps=eps,
                                 nleapfrog=nleapfrog)
