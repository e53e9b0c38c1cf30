"""stheno_amd -- the dense Gaussian-process inference hot path of wesselb/stheno on AMD
Instinct MI355X (gfx950): kernel-matrix construction, Cholesky of K + noise, and the
triangular solves behind ``Normal.logpdf`` / posterior conditioning, as hand-written HIP
kernels (``stheno_amd/csrc``, C ABI in ``include/gpk.h``) behind the ``stheno.torch``
API surface for that path (``stheno/__init__.py:1-28``)."""
from . import B  # noqa: F401
from .kernels import (  # noqa: F401
    EQ, Exp, Kernel, Linear, Matern12, Matern32, Matern52, OneKernel, OneMean, PosteriorKernel,
    PosteriorMean, SubspaceKernel, ZeroKernel, ZeroMean,
)
from .lazy import LazyMatrix, LazyVector  # noqa: F401
from .matrix import Dense, Diagonal, Zero, deferred_checks  # noqa: F401
from .model import *  # noqa: F401,F403
from .random import Normal, Random, RandomProcess, RandomVector  # noqa: F401

__version__ = "0.1.0"


class BreakingChangeWarning(UserWarning):
    """A breaking change."""
