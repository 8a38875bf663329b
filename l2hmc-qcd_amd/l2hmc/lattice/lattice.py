"""``Lattice`` abstract base class -- mirrors src/l2hmc/lattice/lattice.py:20-227."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Optional

import numpy as np

from l2hmc.configs import Charges
from l2hmc.group.group import Group


class Lattice(ABC):
    def __init__(self, group: Group, nchains: int, shape: list[int]) -> None:
        self.g = group
        self.link_shape = self.g._shape
        self.xshape = [self.g._dim, *shape]
        if len(self.g._shape) > 1:
            self.xshape.extend(self.g._shape)
        self.dim = self.g._dim
        self._shape = [nchains, *self.xshape]
        self.nchains = nchains
        self._lattice_shape = list(shape)
        self.volume = int(np.cumprod(shape)[-1])

    def draw_batch(self) -> Any:
        return self.g.random(list(self._shape[:-2]))

    def update_link(self, x: Any, p: Any) -> Any:
        return self.g.update_gauge(x, p)

    def random(self) -> Any:
        return self.g.random(list(self._shape))

    def random_momentum(self) -> Any:
        return self.g.random_momentum(list(self._shape))

    @abstractmethod
    def action(self, x: Any, beta: Any) -> Any: ...

    @abstractmethod
    def kinetic_energy(self, v: Any) -> Any: ...

    def potential_energy(self, x: Any, beta: Any) -> Any:
        return self.action(x, beta)

    @abstractmethod
    def wilson_loops(self, x: Any) -> Any: ...

    @abstractmethod
    def _plaqs(self, wloops: Any) -> Any: ...

    def plaqs(self, x: Optional[Any] = None, wloops: Optional[Any] = None) -> Any:
        if wloops is None:
            assert x is not None
            wloops = self.wilson_loops(x)
        return self._plaqs(wloops)

    def charges(self, x: Optional[Any] = None, wloops: Optional[Any] = None) -> Charges:
        if wloops is None:
            assert x is not None
            wloops = self.wilson_loops(x)
        return self._charges(wloops=wloops)

    @abstractmethod
    def _charges(self, wloops: Any) -> Charges: ...

    def sin_charges(self, x: Optional[Any] = None, wloops: Optional[Any] = None) -> Any:
        if wloops is None:
            assert x is not None
            wloops = self.wilson_loops(x)
        return self._sin_charges(wloops)

    @abstractmethod
    def _sin_charges(self, wloops: Any) -> Any: ...

    def int_charges(self, x: Optional[Any] = None, wloops: Optional[Any] = None) -> Any:
        if wloops is None:
            assert x is not None
            wloops = self.wilson_loops(x)
        return self._int_charges(wloops)

    @abstractmethod
    def _int_charges(self, wloops: Any) -> Any: ...

    def unnormalized_log_prob(self, x: Any, beta: Any) -> Any:
        return self.action(x=x, beta=beta)

    @abstractmethod
    def grad_action(self, x: Any, beta: Any) -> Any: ...

    @abstractmethod
    def action_with_grad(self, x: Any, beta: Any) -> tuple[Any, Any]: ...

    @abstractmethod
    def calc_metrics(self, x: Any, beta: Optional[Any] = None) -> dict[str, Any]: ...
