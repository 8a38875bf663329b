"""Config dataclasses of the hot path -- same names, fields, defaults and derived attributes
as the reference's ``src/l2hmc/configs.py`` (``DynamicsConfig :458-520``, ``NetworkConfig
:437-455``, ``ConvolutionConfig :393-434``, ``NetWeight(s) :278-317``, ``LossConfig :523-538``,
``InputSpec :541-571``, ``Steps :344-390``, ``LearningRateConfig :320-341``,
``AnnealingSchedule :803-873``, ``ExperimentConfig :641-800``, ``Charges :184-187``,
``LatticeMetrics :190-202``, ``State :142-146``).  Unlike the reference, importing this
module creates no directories and needs neither hydra nor omegaconf; ``get_config`` composes
the YAML tree under ``l2hmc/conf`` with the same defaults-list / dotted-override syntax.
"""
from __future__ import annotations

import json
import logging
import os
from abc import ABC, abstractmethod
from copy import deepcopy
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

logger = logging.getLogger(__name__)

HERE = Path(os.path.abspath(__file__)).parent
CONF_DIR = HERE.joinpath('conf')

FP16_SYNONYMS = ['float16', 'fp16', '16', 'half']
BF16_SYNONYMS = ['bfloat16', 'bf16']
FP32_SYNONYMS = ['float32', 'fp32', '32', 'single']
FP64_SYNONYMS = ['float64', 'fp64', '64', 'double']

SYNONYMS = {
    'pytorch': ['p', 'pt', 'torch', 'pytorch'],
    'tensorflow': ['t', 'tf', 'tflow', 'tensorflow'],
    'horovod': ['h', 'hv', 'hvd', 'horovod'],
    'DDP': ['ddp'],
    'deepspeed': ['ds', 'deepspeed'],
}


def dict_to_list_of_overrides(d: dict):
    return [f'{k}={v}' for k, v in flatten_dict(d, sep='.').items()]


def flatten_dict(d: dict, sep: str = '/', pre='') -> dict:
    return {
        pre + sep + k if pre else k: v
        for kk, vv in d.items()
        for k, v in flatten_dict(vv, sep, kk).items()
    } if isinstance(d, dict) else {pre: d}


def list_to_str(x: list) -> str:
    if isinstance(x[0], int):
        return '-'.join([str(int(i)) for i in x])
    if isinstance(x[0], float):
        return '-'.join([f'{i:2.1f}' for i in x])
    return '-'.join([str(i) for i in x])


@dataclass
class State:
    x: Any
    v: Any
    beta: Any


@dataclass
class BaseConfig(ABC):
    @abstractmethod
    def to_str(self) -> str:
        pass

    def to_json(self) -> str:
        return json.dumps(self.__dict__, default=str)

    def get_config(self) -> dict:
        return asdict(self)

    def asdict(self) -> dict:
        return asdict(self)

    def to_dict(self) -> dict:
        return deepcopy(self.__dict__)

    def to_file(self, fpath) -> None:
        """the JSON text of the config, as a JSON string (configs.py:169-171 dumps `to_json()`'s string)"""
        with open(fpath, 'w') as f:
            json.dump(self.to_json(), f, indent=4)

    def from_file(self, fpath) -> None:
        """re-initialise from a file written by `to_file`.  (The reference's version opens the file for
        writing first and so truncates it, configs.py:173-178; this one reads it.)"""
        with open(fpath, 'r') as f:
            config = json.load(f)
        if isinstance(config, str):
            config = json.loads(config)
        fields = getattr(self, '__dataclass_fields__', None)
        if fields is not None:
            config = {k: v for k, v in config.items() if k in fields and fields[k].init}
        self.__init__(**config)

    def __getitem__(self, key):
        return super().__getattribute__(key)


@dataclass
class Charges:
    intQ: Any
    sinQ: Any


@dataclass
class LatticeMetrics:
    plaqs: Any
    charges: Charges
    p4x4: Any

    def asdict(self) -> dict:
        return {'plaqs': self.plaqs, 'sinQ': self.charges.sinQ, 'intQ': self.charges.intQ,
                'p4x4': self.p4x4}


@dataclass
class NetWeight(BaseConfig):
    """Scales the (s, t, q) network functions."""
    s: float = field(default=1.)
    t: float = field(default=1.)
    q: float = field(default=1.)

    def to_dict(self):
        return {'s': self.s, 't': self.t, 'q': self.q}

    def to_str(self):
        return f's{self.s:2.1f}t{self.t:2.1f}q{self.t:2.1f}'


@dataclass
class NetWeights(BaseConfig):
    x: NetWeight = field(default_factory=lambda: NetWeight(1., 1., 1.))
    v: NetWeight = field(default_factory=lambda: NetWeight(1., 1., 1.))

    def to_str(self):
        return f'nwx-{self.x.to_str()}-nwv-{self.v.to_str()}'

    def to_dict(self):
        return {'x': self.x.to_dict(), 'v': self.v.to_dict()}

    def __post_init__(self):
        if not isinstance(self.x, NetWeight):
            self.x = NetWeight(**self.x)
        if not isinstance(self.v, NetWeight):
            self.v = NetWeight(**self.v)


@dataclass
class LearningRateConfig(BaseConfig):
    lr_init: float = 1e-3
    mode: str = 'auto'
    monitor: str = 'loss'
    patience: int = 5
    cooldown: int = 0
    warmup: int = 1000
    verbose: bool = True
    min_lr: float = 1e-6
    factor: float = 0.98
    min_delta: float = 1e-4
    clip_norm: float = 2.0

    def to_str(self):
        return f'lr-{self.lr_init:3.2f}'


@dataclass
class Steps(BaseConfig):
    nera: int
    nepoch: int
    test: int
    log: int = 100
    print: int = 200
    extend_last_era: Optional[int] = None

    def __post_init__(self):
        if self.extend_last_era is None:
            self.extend_last_era = 1
        self.total = self.nera * self.nepoch
        freq = int(self.nepoch // 20)
        self.log = max(1, freq) if self.log is None else self.log
        self.print = max(1, freq) if self.print is None else self.print

    def to_str(self) -> str:
        return f'nera-{self.nera}_nepoch-{self.nepoch}'

    def update(self, nera: Optional[int] = None, nepoch: Optional[int] = None, test: Optional[int] = None,
               log: Optional[int] = None, print: Optional[int] = None,
               extend_last_era: Optional[int] = None) -> 'Steps':
        """a copy with the given fields replaced (configs.py:371-388)"""
        pick = lambda new, old: old if new is None else new
        return Steps(nera=pick(nera, self.nera), nepoch=pick(nepoch, self.nepoch), test=pick(test, self.test),
                     log=pick(log, self.log), print=pick(print, self.print),
                     extend_last_era=pick(extend_last_era, self.extend_last_era))


@dataclass
class ConvolutionConfig(BaseConfig):
    filters: Optional[Sequence[int]] = None
    sizes: Optional[Sequence[int]] = None
    pool: Optional[Sequence[int]] = None

    def __post_init__(self):
        if self.filters is None:
            return
        if self.sizes is None:
            logger.warning('Using default filter size of 2')
            self.sizes = list(len(self.filters) * [2])
        if self.pool is None:
            logger.warning('Using default pooling size of 2')
            self.pool = len(self.filters) * [2]
        assert len(self.filters) == len(self.sizes)
        assert len(self.filters) == len(self.pool)

    def to_str(self) -> str:
        if self.filters is None:
            return 'conv-None'
        if len(self.filters) > 0:
            outstr = [list_to_str(list(self.filters))]
            if self.sizes is not None:
                outstr.append(list_to_str(list(self.sizes)))
            if self.pool is not None:
                outstr.append(list_to_str(list(self.pool)))
            return '-'.join(['conv', '_'.join(outstr)])
        return ''


@dataclass
class NetworkConfig(BaseConfig):
    units: Sequence[int]
    activation_fn: str
    dropout_prob: float
    use_batch_norm: bool = True

    def to_str(self):
        ustr = '-'.join([str(int(i)) for i in self.units])
        return '-'.join(['net', '_'.join([ustr, f'dp-{self.dropout_prob:2.1f}',
                                          f'bn-{self.use_batch_norm}'])])


@dataclass
class DynamicsConfig(BaseConfig):
    nchains: int
    group: str
    latvolume: List[int]
    nleapfrog: int
    eps: float = 0.01
    eps_hmc: float = 0.01
    use_ncp: bool = True
    verbose: bool = True
    eps_fixed: bool = False
    use_split_xnets: bool = True
    use_separate_networks: bool = True
    merge_directions: bool = True

    def to_str(self) -> str:
        latstr = '-'.join([str(i) for i in self.xshape[1:]])
        return '/'.join([self.group, latstr, f'nlf-{self.nleapfrog}',
                         f'xsplit-{self.use_split_xnets}',
                         f'sepnets-{self.use_separate_networks}',
                         f'merge-{self.merge_directions}'])

    def __post_init__(self):
        assert self.group.upper() in ['U1', 'SU3']
        if self.eps_hmc is None:
            self.eps_hmc = 1.0 / self.nleapfrog          # trajectory length 1
        self.latvolume = [int(i) for i in self.latvolume]
        if self.group.upper() == 'U1':
            self.dim = 2
            self.nt, self.nx = self.latvolume
            self.xshape = (self.nchains, self.dim, *self.latvolume)
            self.vshape = (self.nchains, self.dim, *self.latvolume)
            assert len(self.latvolume) == 2
        else:
            self.dim = 4
            self.link_shape = (3, 3)
            self.vec_shape = 8
            self.nt, self.nx, self.ny, self.nz = self.latvolume
            self.xshape = (self.nchains, self.dim, *self.latvolume, *self.link_shape)
            self.vshape = (self.nchains, self.dim, *self.latvolume, self.vec_shape)
            assert len(self.latvolume) == 4
        self.xdim = int(np.cumprod(self.xshape[1:])[-1])


@dataclass
class LossConfig(BaseConfig):
    use_mixed_loss: bool = False
    charge_weight: float = 0.01
    rmse_weight: float = 0.0
    plaq_weight: float = 0.0
    aux_weight: float = 0.0

    def to_str(self) -> str:
        return '_'.join([f'qw-{self.charge_weight:2.1f}', f'pw-{self.plaq_weight:2.1f}',
                         f'rw-{self.rmse_weight:2.1f}', f'aw-{self.aux_weight:2.1f}',
                         f'mixed-{self.use_mixed_loss}'])


@dataclass
class InputSpec(BaseConfig):
    xshape: Sequence[int]
    xnet: Optional[Dict[str, Any]] = None
    vnet: Optional[Dict[str, Any]] = None

    def to_str(self):
        return '-'.join([str(i) for i in self.xshape])

    def __post_init__(self):
        if len(self.xshape) == 2:
            self.xdim = self.xshape[-1]
            self.vshape = self.xshape
            self.vdim = self.xshape[-1]
        elif len(self.xshape) > 2:
            self.xdim = int(np.cumprod(self.xshape[1:])[-1])
            lat_shape = self.xshape[:-2]
            vd = (self.xshape[-1] ** 2) - 1
            self.vshape = (*lat_shape, vd)
            self.vdim = int(np.cumprod(self.vshape[1:])[-1])
        else:
            raise ValueError(f'Invalid `xshape`: {self.xshape}')
        if self.xnet is None:
            self.xnet = {'x': self.xshape, 'v': self.xshape}
        if self.vnet is None:
            self.vnet = {'x': self.xshape, 'v': self.xshape}


@dataclass
class AnnealingSchedule(BaseConfig):
    beta_init: float
    beta_final: Optional[float] = 1.0
    dynamic: bool = False

    def to_str(self) -> str:
        return f'bi-{self.beta_init}_bf-{self.beta_final}'

    def __post_init__(self):
        if self.beta_final is None or self.beta_final < self.beta_init:
            self.beta_final = float(self.beta_init)
        self.beta_init = float(self.beta_init)
        self.beta_final = float(self.beta_final)

    def update(self, beta_init: Optional[float] = None, beta_final: Optional[float] = None):
        if beta_init is not None:
            self.beta_init = beta_init
        if beta_final is not None:
            self.beta_final = beta_final

    def setup(self, nera: Optional[int] = None, nepoch: Optional[int] = None,
              steps: Optional[Steps] = None, beta_init: Optional[float] = None,
              beta_final: Optional[float] = None) -> dict:
        if nera is None:
            assert steps is not None
            nera = steps.nera
        if nepoch is None:
            assert steps is not None
            nepoch = steps.nepoch
        beta_init = self.beta_init if beta_init is None else beta_init
        if beta_final is None:
            beta_final = self.beta_final if self.beta_final is not None else self.beta_init
        self.betas = np.linspace(beta_init, beta_final, nera)
        total = steps.total if steps is not None else 1
        self._dbeta = (beta_final - beta_init) / total
        self.beta_dict = {str(era): self.betas[era] for era in range(nera)}
        return self.beta_dict


@dataclass
class ExperimentConfig(BaseConfig):
    """Top-level config (configs.py:641-800).  Tracking back-ends (wandb / aim / deepspeed /
    horovod) are not part of this build; their flags are accepted and ignored."""
    wandb: Any
    steps: Steps
    framework: str
    loss: LossConfig
    network: NetworkConfig
    conv: ConvolutionConfig
    net_weights: NetWeights
    dynamics: DynamicsConfig
    learning_rate: LearningRateConfig
    annealing_schedule: AnnealingSchedule
    gradient_accumulation_steps: int = 1
    restore: bool = True
    save: bool = True
    c1: float = 0.0
    port: str = '2345'
    compile: bool = True
    profile: bool = False
    init_aim: bool = True
    init_wandb: bool = True
    use_wandb: bool = True
    use_tb: bool = False
    debug_mode: bool = False
    default_mode: bool = True
    print_config: bool = True
    precision: str = 'float32'
    ignore_warnings: bool = True
    backend: str = 'DDP'
    seed: Optional[int] = None
    ds_config_path: Optional[Any] = None
    name: Optional[str] = None
    width: Optional[int] = None
    nchains: Optional[int] = None
    compression: Optional[str] = None

    def __post_init__(self):
        if self.seed is None:
            self.seed = 0
        self.xdim = self.dynamics.xdim
        self.xshape = self.dynamics.xshape
        self.micro_batch_size = self.dynamics.nchains
        world = int(os.environ.get('WORLD_SIZE', 1))
        self.global_batch_size = world * self.micro_batch_size * self.gradient_accumulation_steps
        p = str(self.precision)
        if p in FP16_SYNONYMS:
            self.precision = 'fp16'
        elif p in BF16_SYNONYMS:
            self.precision = 'bf16'
        elif p in FP32_SYNONYMS:
            self.precision = 'float32'
        elif p in FP64_SYNONYMS:
            self.precision = 'float64'
        if self.framework not in SYNONYMS['pytorch']:
            raise ValueError('this build implements the PyTorch path only '
                             f'(framework={self.framework!r})')
        self.annealing_schedule.setup(nera=self.steps.nera, nepoch=self.steps.nepoch)

    def to_str(self) -> str:
        return '/'.join([self.dynamics.to_str(), self.conv.to_str(), self.network.to_str(),
                         self.framework])


def get_config(overrides: Optional[list[str]] = None, config_name: str = 'config') -> dict:
    """Compose ``conf/config.yaml`` + group defaults + dotted overrides (hydra-compatible
    subset: defaults list, ``group=option``, ``a.b.c=value``, ``${a.b}`` interpolation).
    ``config_name`` other than 'config' names a FLAT primary file (conf/su3test.yaml,
    conf/su3-min.yaml of the reference): it is layered over the composed defaults, command-line
    overrides on top."""
    from l2hmc.utils.compose import compose, compose_flat
    if config_name in (None, 'config'):
        return compose(CONF_DIR, 'config', overrides or [])
    return compose_flat(CONF_DIR, config_name, overrides or [])


def get_experiment(overrides: Optional[list[str]] = None, build_networks: bool = True,
                   keep: Optional[str | list[str]] = None,
                   skip: Optional[str | list[str]] = None):
    """Compose the config and build the Experiment (configs.py:1008-1034 of the reference; the
    entry its own SU(3) smoke script uses, train4dSU3.py:207).  PyTorch only: the TensorFlow
    back-end is outside this build."""
    cfg = get_config(overrides)
    framework = str(cfg.get('framework'))
    if framework in SYNONYMS['pytorch']:
        from l2hmc.experiment.pytorch.experiment import Experiment
        return Experiment(cfg, keep=keep, skip=skip, build_networks=build_networks)
    if framework in SYNONYMS['tensorflow']:
        raise ValueError('get_experiment: framework=tensorflow is not part of this build '
                         '(the hot path is the PyTorch one)')
    raise ValueError(f'Unexpected value for `cfg.framework: {framework}')


def instantiate(cfg: dict) -> ExperimentConfig:
    from l2hmc.utils.compose import instantiate as _inst
    return _inst(cfg)
