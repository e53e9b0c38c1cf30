"""The few ``lab`` (``B.*``) names user code of the reference touches on this path
(``from stheno import B`` / ``import lab as B``): the global jitter ``B.epsilon``,
``B.dense`` and some tensor constructors that default to the HIP device.

``B.epsilon`` is read at factorisation time, like in the reference
(``README.md:820-831``): ``B.epsilon = 1e-6`` before an fp32 computation.
"""
import sys
import types

import torch

from . import matrix as _matrix

__all__ = ["epsilon", "dense", "to_numpy", "default_device", "linspace", "randn", "rand", "zeros", "ones", "eye"]

default_dtype = torch.float64


def default_device():
    """``cuda`` (HIP) when a GPU is visible, else ``cpu`` (host-logic tests only)."""
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def dense(a):
    """Strip matrix structure: a plain ``torch.Tensor``."""
    if isinstance(a, _matrix.AbstractMatrix):
        return a.dense()
    return a


def to_numpy(a):
    return dense(a).detach().cpu().numpy()


def _kw(dtype, device):
    return dict(dtype=dtype or default_dtype, device=device or default_device())


def linspace(a, b, n, dtype=None, device=None):
    return torch.linspace(a, b, n, **_kw(dtype, device))


def randn(*shape, dtype=None, device=None, generator=None):
    return torch.randn(*shape, generator=generator, **_kw(dtype, device))


def rand(*shape, dtype=None, device=None, generator=None):
    return torch.rand(*shape, generator=generator, **_kw(dtype, device))


def zeros(*shape, dtype=None, device=None):
    return torch.zeros(*shape, **_kw(dtype, device))


def ones(*shape, dtype=None, device=None):
    return torch.ones(*shape, **_kw(dtype, device))


def eye(n, dtype=None, device=None):
    return torch.eye(n, **_kw(dtype, device))


class _BModule(types.ModuleType):
    """Module whose ``epsilon`` attribute is the library-wide jitter."""

    @property
    def epsilon(self):
        return _matrix.config.epsilon

    @epsilon.setter
    def epsilon(self, value):
        _matrix.config.epsilon = float(value)

    @property
    def check_info(self):
        return _matrix.config.check_info

    @check_info.setter
    def check_info(self, value):
        _matrix.config.check_info = bool(value)


sys.modules[__name__].__class__ = _BModule
