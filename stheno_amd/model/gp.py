"""``GP``: a handle on a process registered in a ``Measure`` (``stheno/model/gp.py``)."""
import weakref
from types import FunctionType

import torch

from .. import kernels as _k
from ..random import RandomProcess
from .fdd import FDD

__all__ = ["GP", "cross", "assert_same_measure", "intersection_measure_group"]


def assert_same_measure(*ps):
    for p in ps[1:]:
        if ps[0].measure != p.measure:
            raise AssertionError(f"Processes {ps[0]} and {p} are associated to different measures.")


def intersection_measure_group(*ps):
    assert_same_measure(*ps)
    inter = set(ps[0]._measures)
    for p in ps[1:]:
        inter &= set(p._measures)
    return inter


def cross(*ps):
    """Cartesian product of processes (``gp.py:43-55``): a process over multi-process inputs
    ``(f1(x1), f2(x2), ...)``, used to observe / sample several processes jointly."""
    p_cross = GP()
    p_cross._parents = tuple(ps)
    for measure in intersection_measure_group(*ps):
        measure.cross(p_cross, *ps)
    return p_cross


def _is_numeric(v):
    return isinstance(v, (int, float)) or torch.is_tensor(v)


class GP(RandomProcess):
    """``GP(kernel)``, ``GP(mean, kernel)``; keyword arguments ``measure`` and ``name``
    (``gp.py:70-103``).  ``GP()`` creates an unattached handle."""

    def __init__(self, *args, measure=None, name=None):
        from .measure import Measure

        # The measure the GP was created under is held strongly; measures that merely know the
        # GP (posteriors conditioned later) are back-referenced WEAKLY, so an old posterior --
        # and the Cholesky factor it owns -- dies with its last user handle, not with the prior GP.
        self._measure0 = None
        self._backrefs = weakref.WeakSet()
        self._parents = ()      # processes this one is derived from (kept alive: rules refer to their ids)
        if len(args) == 0:
            return
        if len(args) == 1:
            mean, kernel = _k.ZeroMean(), args[0]
        elif len(args) == 2:
            mean, kernel = args
        else:
            raise TypeError("GP(kernel) or GP(mean, kernel)")
        if measure is None:
            measure = Measure.default if Measure.default else Measure()
        if _is_numeric(mean) or isinstance(mean, FunctionType):
            mean = _k._wrap_mean(mean)
        if _is_numeric(kernel):
            kernel = kernel * _k.OneKernel()
        measure.add_independent_gp(self, mean, kernel)
        if name:
            measure.name(self, name)

    @property
    def _measures(self):
        return ([self._measure0] if self._measure0 is not None else []) + list(self._backrefs)

    def _attach(self, measure):
        if self._measure0 is None:
            self._measure0 = measure
        elif measure is not self._measure0:
            self._backrefs.add(measure)

    @property
    def measure(self):
        if self._measure0 is None:
            raise RuntimeError("GP is not associated to a measure.")
        return self._measure0

    @property
    def kernel(self):
        return self.measure.kernels[self]

    @property
    def mean(self):
        return self.measure.means[self]

    @property
    def name(self):
        return self.measure[self]

    @name.setter
    def name(self, name):
        for measure in self._measures:
            measure.name(self, name)

    def __call__(self, x, noise=None):
        """Finite-dimensional distribution at ``x`` (``gp.py:134-144``)."""
        return FDD(self, x, noise)

    def condition(self, *args):
        """Condition ``self.measure`` on data and return the posterior GP (``gp.py:146-155``)."""
        return self.measure.condition(*args)(self)

    def __or__(self, args):
        return self.condition(*args) if isinstance(args, tuple) else self.condition(args)

    # -- the bookkeeping-only part of the GP algebra -----------------------------
    def __add__(self, other):
        res = GP()
        if isinstance(other, GP):
            res._parents = (self, other)
            for measure in intersection_measure_group(self, other):
                measure.sum(res, self, other)
        else:
            res._parents = (self,)
            for measure in self._measures:
                measure.sum(res, self, other)
        return res

    def __mul__(self, other):
        if isinstance(other, GP) or isinstance(other, FunctionType):
            raise NotImplementedError("products with processes/functions are outside the accelerated path")
        res = GP()
        res._parents = (self,)
        for measure in self._measures:
            measure.mul(res, self, other)
        return res

    @property
    def stationary(self):
        return self.kernel.stationary

    def __repr__(self):
        return f"GP({self.mean!r}, {self.kernel!r})" if self._measures else "GP()"

    __str__ = __repr__
