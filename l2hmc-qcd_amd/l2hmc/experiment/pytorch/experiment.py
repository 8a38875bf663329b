"""``Experiment`` -- API shape of src/l2hmc/experiment/pytorch/experiment.py:141-450
(``Experiment(cfg).evaluate(job_type)``, ``.trainer``) without the tracking back-ends."""
from __future__ import annotations

from typing import Optional

import torch

import l2hmc.configs as cfgs
from l2hmc.trainers.pytorch.trainer import Trainer
from l2hmc.utils import dist as D


class Experiment:
    def __init__(self, cfg: dict | cfgs.ExperimentConfig, build_networks: bool = True,
                 keep=None, skip=None, seed: Optional[int | bool] = None) -> None:
        """seed (not a reference argument): an int, or True for `cfg.seed`, seeds torch / numpy /
        random right here -- for a direct caller (a script, a test) that wants `Experiment(cfg)` alone to be
        reproducible.  None keeps the reference's contract (the caller has seeded); a generator that was never
        seeded in this process is reported once, because such a run cannot be reproduced from `cfg.seed`."""
        self.cfg = cfg
        self.config = cfgs.instantiate(cfg) if isinstance(cfg, dict) else cfg
        if seed is not None and seed is not False:
            D.seed_everything(int(self.config.seed if seed is True else seed))
        elif not D.SEEDED[0] and not getattr(Experiment, '_warned_unseeded', False):
            import logging
            logging.getLogger('l2hmc').info(
                'Experiment: the host generator has not been seeded through l2hmc.utils.dist (seed_everything / '
                'setup_torch); like the reference, Experiment does not seed -- pass seed=True to use cfg.seed')
            Experiment._warned_unseeded = True
        # Seeding is the CALLER's job, like in the reference: its Experiment never seeds
        # (experiment/pytorch/experiment.py:141-225); `python -m l2hmc` seeds with cfg.seed through
        # setup_torch right before it builds the Experiment (__main__.py:78-92) and a script such
        # as train4dSU3.py seeds with its own value (:61) before configs.get_experiment.  The
        # networks and masks drawn below therefore continue the caller's generator stream exactly
        # as the reference's do, and so does everything after construction (start configuration,
        # momenta, accept uniforms): one process, same seed -> the reference's CPU-path chain.
        #
        # Data parallelism: the MODEL must be identical on every rank, the CHAINS
        # must differ.  Rank 0's parameters / buffers and masks are broadcast (DDP constructor
        # semantics, trainers/pytorch/trainer.py:246-257 of the reference; the masks too, which the
        # reference leaves rank-dependent) and ranks > 0 then move to their own chain streams
        # (`chain_seed`); rank 0 keeps its stream, i.e. stays the single-process chain.
        env = D.setup_torch_distributed(self.config.backend, self.config.port)
        self._rank = env['rank']
        self.trainer = Trainer(self.config, build_networks=build_networks, keep=keep, skip=skip)
        self.lattice = self.trainer.lattice
        D.sync_model(self.trainer.dynamics)
        if env['world_size'] > 1 and self._rank > 0:
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
