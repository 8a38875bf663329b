"""``Group`` abstract base class -- mirrors src/l2hmc/group/group.py:22-81."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Optional, Sequence


class Group(ABC):
    """Gauge group represented as matrices in the last two dimensions."""

    def __init__(self, dim: int, shape: Sequence[int], dtype: Any,
                 name: Optional[str] = None) -> None:
        self._dim = dim
        self._shape = shape
        self._dtype = dtype
        if name is not None:
            self._name = name

    @abstractmethod
    def exp(self, x: Any) -> Any: ...

    @abstractmethod
    def mul(self, a: Any, b: Any, adjoint_a: bool = False, adjoint_b: bool = False) -> Any: ...

    @abstractmethod
    def update_gauge(self, x: Any, p: Any) -> Any: ...

    @abstractmethod
    def adjoint(self, x: Any) -> Any: ...

    @abstractmethod
    def trace(self, x: Any) -> Any: ...

    @abstractmethod
    def compat_proj(self, x: Any) -> Any: ...

    @abstractmethod
    def random(self, shape: list[int]) -> Any: ...

    @abstractmethod
    def random_momentum(self, shape: list[int]) -> Any: ...

    @abstractmethod
    def kinetic_energy(self, p: Any) -> Any: ...
