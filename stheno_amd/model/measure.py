"""``Measure``: a joint model over processes (``stheno/model/measure.py``), reduced to the
bookkeeping the dense inference path needs: registering processes, lazy (cross-)kernels
and means, conditioning, re-binding a GP / FDD under a posterior, ``logpdf`` and ``sample``."""
import weakref

import torch

from .. import kernels as _k
from ..lazy import LazyMatrix, LazyVector
from .fdd import FDD
from .gp import GP, assert_same_measure
from .observations import AbstractObservations, AbstractPseudoObservations, Observations, combine

__all__ = ["Measure"]


class Measure:
    """A GP model.  ``with Measure() as prior:`` makes it the default for new GPs
    (``measure.py:35-55``)."""

    default = None

    def __init__(self):
        # Processes are held WEAKLY (a GP holds its measure strongly): no GP <-> Measure cycle,
        # so dropping the user's handles frees a posterior and its factor immediately.  When a
        # process dies its id is purged from this measure (ids may be recycled by CPython).
        self._ps = []
        self._pids = set()
        self.means = LazyVector()
        self.kernels = LazyMatrix()
        self._gps_by_name = {}
        self._names_by_gp = {}
        self._prev_default = None

    @property
    def ps(self):
        """The (live) processes of the measure."""
        return [p for p in (r() for r in self._ps) if p is not None]

    @staticmethod
    def _purge(wself, pid):
        self = wself()
        if self is not None:
            self._pids.discard(pid)
            self.means.purge(pid)
            self.kernels.purge(pid)
            name = self._names_by_gp.pop(pid, None)
            if name is not None:
                self._gps_by_name.pop(name, None)

    def _register(self, p):
        self._ps.append(weakref.ref(p))
        self._pids.add(id(p))
        p._attach(self)
        weakref.finalize(p, Measure._purge, weakref.ref(self), id(p))

    def __enter__(self):
        self._prev_default = Measure.default
        Measure.default = self
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        Measure.default = self._prev_default

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other

    # -- naming (measure.py:62-91) ----------------------------------------------
    def __getitem__(self, key):
        if isinstance(key, str):
            return self._gps_by_name[key]
        return self._names_by_gp[id(key)]

    def name(self, p, name):
        if id(p) in self._names_by_gp:
            del self._gps_by_name[self._names_by_gp[id(p)]]
            del self._names_by_gp[id(p)]
        if name in self._gps_by_name:
            raise RuntimeError(f'Name "{name}" for "{p}" already taken by "{self[name]}".')
        self._gps_by_name[name] = p
        self._names_by_gp[id(p)] = name

    # -- registering processes ---------------------------------------------------
    def _add_p(self, p):
        self._register(p)

    def _update(self, p, mean, kernel, left_rule, right_rule=None):
        # rules live inside this measure's own stores: they refer to it weakly and to processes
        # by id, or the measure would be a reference cycle on its own
        self.means[p] = mean
        self.kernels[p] = kernel
        wself, pid = weakref.ref(self), id(p)
        self.kernels.add_left_rule(pid, self._pids, left_rule)
        if right_rule:
            self.kernels.add_right_rule(pid, self._pids, right_rule)
        else:
            self.kernels.add_right_rule(pid, self._pids, lambda i: reversed(wself().kernels[pid, i]))
        self._add_p(p)
        return p

    def add_independent_gp(self, p, mean, kernel):
        """Register ``p`` with zero cross-covariance to every other process (``measure.py:156-178``)."""
        self.means[p] = mean
        self.kernels[p] = kernel
        self.kernels.add_left_rule(id(p), self._pids, lambda j: _k.ZeroKernel())
        self.kernels.add_right_rule(id(p), self._pids, lambda i: _k.ZeroKernel())
        self._add_p(p)
        return p

    def __call__(self, p):
        """``measure(p)``: a new GP that is ``p`` under this measure; ``measure(fdd)``: the FDD
        re-bound under this measure (``measure.py:139-154``)."""
        if isinstance(p, FDD):
            return self(p.p)(p.x, p.noise)
        p_copy = GP()
        p_copy._parents = (p,)
        wself, pid = weakref.ref(self), id(p)
        return self._update(p_copy, self.means[p], self.kernels[p], lambda j: wself().kernels[pid, j],
                            lambda i: wself().kernels[i, pid])

    # -- bookkeeping-only algebra (measure.py:180-239) ---------------------------
    def sum(self, p_sum, p, other):
        if isinstance(other, GP):
            assert_same_measure(p, other)
            p1, p2 = p, other
            wself, i1, i2 = weakref.ref(self), id(p1), id(p2)
            return self._update(
                p_sum, self.means[p1] + self.means[p2],
                self.kernels[p1] + self.kernels[p2] + self.kernels[p1, p2] + self.kernels[p2, p1],
                lambda j: wself().kernels[i1, j] + wself().kernels[i2, j],
            )
        if torch.is_tensor(other) and other.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("a learnable constant added to a process would be detached here: pass it as the mean")
        wself, pid = weakref.ref(self), id(p)
        return self._update(p_sum, self.means[p] + other, self.kernels[p], lambda j: wself().kernels[pid, j])

    def mul(self, p_mul, p, other):
        if torch.is_tensor(other) and other.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("a learnable scale of a process would be detached here: put it on the kernel "
                                      "(`GP(c**2 * EQ())`), where variances are differentiable")
        v = float(other)
        wself, pid = weakref.ref(self), id(p)
        return self._update(p_mul, self.means[p] * v, self.kernels[p] * v**2, lambda j: wself().kernels[pid, j] * v)

    def cross(self, p_cross, *ps):
        """Register ``p_cross`` as the Cartesian product of ``ps`` (``measure.py:404-423``): its
        kernel / mean are the block kernel matrix / stacked mean over multi-process inputs."""
        mok = _k.MultiOutputKernel(self, ps)
        return self._update(p_cross, _k.MultiOutputMean(self, ps), mok, lambda j: _k._CrossKernel(mok, j))

    # -- conditioning (measure.py:362-401) --------------------------------------
    def condition(self, *args):
        if len(args) == 1 and isinstance(args[0], AbstractObservations):
            obs = args[0]
        elif len(args) == 2 and isinstance(args[0], FDD):
            obs = Observations(*args)
        elif len(args) == 1 and isinstance(args[0], tuple) and len(args[0]) == 2 and isinstance(args[0][0], FDD):
            obs = Observations(*args[0])
        elif len(args) == 1 and isinstance(args[0], tuple):
            obs = Observations(*args[0])
        else:
            obs = Observations(*args)
        posterior = Measure()
        live = self.ps
        posterior._pids = {id(p) for p in live}
        posterior.means.add_rule(posterior._pids, lambda i: obs.posterior_mean(self, i))
        posterior.kernels.add_rule(posterior._pids, lambda i, j: obs.posterior_kernel(self, i, j))
        for p in live:
            posterior._register(p)          # (weak) back-reference, measure.py:381-383
        posterior._conditioned_on = (self, obs)      # for the chain-rule form of differentiable posterior log-densities (fdd.py)
        return posterior

    def __or__(self, args):
        return self.condition(*args) if isinstance(args, tuple) and not isinstance(args[0], FDD) else self.condition(args)

    # -- sampling / logpdf (measure.py:425-489) -----------------------------------
    def sample(self, *args, generator=None, xi=None):
        n = 1
        if args and isinstance(args[0], int):
            n, args = args[0], args[1:]
        fdd = combine(*args)
        sample = self(fdd).sample(n, generator=generator, xi=xi)
        if len(args) == 1:
            return sample
        # several FDDs are sampled jointly and handed back one by one (measure.py:440-447)
        out, i = [], 0
        for a in args:
            m = _k.num_elements(a.x)
            out.append(sample[..., i:i + m, :])
            i += m
        return tuple(out)

    def logpdf(self, *args):
        if len(args) == 1 and isinstance(args[0], AbstractPseudoObservations):
            return args[0].elbo(self)
        if len(args) == 1 and isinstance(args[0], Observations):
            return self.logpdf(args[0].fdd, args[0].y)
        if len(args) == 2 and isinstance(args[0], FDD):
            fdd, y = args
            return self(fdd).logpdf(y)
        fdd, y = combine(*args)
        return self(fdd).logpdf(y)
