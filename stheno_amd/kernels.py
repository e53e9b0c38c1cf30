"""Kernels and means on the GP hot path -- the slice of the reference's ``mlkernels``
dependency that ``stheno/model/*.py`` uses:

* primitives ``EQ``, ``Exp``/``Matern12``, ``Matern32``, ``Matern52``, ``Linear``,
  ``OneKernel``, ``ZeroKernel`` with the ``v * k``, ``k1 + k2``, ``k.stretch(l)`` algebra
  (usage: ``readme_example13_optimisation_torch.py:34``, ``tests/model/test_cases.py:138``);
* ``pairwise`` (``k(x, y)``) and ``elwise`` (``k.elwise(x)``) evaluation -- one fused HIP
  kernel launch per call for any sum of primitives (``gpk_kmat`` / ``gpk_kdiag``);
* ``PosteriorKernel``, ``PosteriorMean``, ``SubspaceKernel`` (constructed at
  ``stheno/model/observations.py:148-168,256-277``) and ``mean_var`` / ``mean_var_diag``
  (``stheno/model/fdd.py:69,73``), which share one ``L^{-1} k(z, x)`` between the mean and
  the variance and never form the N* x N* covariance for marginals.

Inputs follow the reference's conventions: ``(N,)`` -> ``(N, 1)``; ``(N, D)``;
``(B, N, D)`` batched.
"""
import numpy as np
import torch

from . import ops
from .matrix import Dense, KernelDense

__all__ = [
    "Kernel", "EQ", "Exp", "Matern12", "Matern32", "Matern52", "Linear", "OneKernel", "ZeroKernel",
    "Mean", "ZeroMean", "OneMean", "PosteriorKernel", "PosteriorMean", "SubspaceKernel",
    "mean_var", "mean_var_diag", "uprank", "num_elements",
    "MultiInput", "MultiOutputKernel", "MultiOutputMean", "InputScaled",
]


class MultiInput:
    """Inputs of a Cartesian product of processes (``cross(f1, f2)((f1(x1), f2(x2)))``): the
    role of the reference's tuple-of-FDDs input (``stheno/mo/input.py:7-36``,
    ``stheno/model/observations.py:28-47``).  ``parts`` is a list of ``(process, x_i)`` with
    ``x_i`` upranked; the object quacks like an ``(N_1 + ... + N_k, .)`` input where the host
    code only asks for ``dtype`` / ``device`` / the number of rows."""

    def __init__(self, parts):
        self.parts = [(p, uprank(x)) for p, x in parts]
        if not self.parts:
            raise ValueError("a multi-input needs at least one part")
        if len({tuple(x.shape[:-2]) for _, x in self.parts}) != 1:
            raise ValueError("the parts of a multi-process input must share their batch dimensions")

    dtype = property(lambda self: self.parts[0][1].dtype)
    device = property(lambda self: self.parts[0][1].device)
    # rows each part contributes: a plain input under a product process stands for ALL its components
    sizes = property(lambda self: [p.kernel.num_outputs(x) for p, x in self.parts])
    shape = property(lambda self: tuple(self.parts[0][1].shape[:-2]) + (sum(self.sizes), 1))

    def dim(self):
        return self.parts[0][1].dim()

    def offsets(self):
        out, o = [], 0
        for n in self.sizes:
            out.append((o, o + n))
            o += n
        return out

    def __getitem__(self, idx):
        """Row selection by a sorted index vector (``B.take`` on the observations, ``fdd.py:125-132``)."""
        if not (torch.is_tensor(idx) and idx.dim() == 1 and idx.dtype == torch.long):
            raise TypeError("a multi-input is sub-selected by an index vector")
        parts = []
        for (p, x), (a, b) in zip(self.parts, self.offsets()):
            sel = idx[(idx >= a) & (idx < b)] - a
            parts.append((p, x[sel]))
        return MultiInput(parts)

    def __repr__(self):
        return "MultiInput(" + ", ".join(f"{tuple(x.shape)}" for _, x in self.parts) + ")"


def uprank(x):
    """``B.uprank``: scalars and vectors become column matrices."""
    if isinstance(x, MultiInput):
        return x
    if not torch.is_tensor(x):
        # host data (NumPy arrays, lists, Python numbers -- what the reference's default backend is fed): one copy to the
        # device the HIP backend computes on; tensors stay where the caller put them (a CPU tensor is refused by the backend)
        x = torch.as_tensor(x)
        if ops.get_backend().name == "hip":
            x = x.to(torch.device("cuda", torch.cuda.current_device()))
    if x.dim() == 0:
        return x.reshape(1, 1)
    if x.dim() == 1:
        return x[:, None]
    return x


def num_elements(x):
    """``mlkernels.num_elements``: number of inputs (rows)."""
    x = uprank(x)
    return x.shape[-2]


def _as_float(v):
    if torch.is_tensor(v):
        if v.numel() != 1:
            raise ValueError("kernel hyper-parameters must be scalars")
        return float(v.detach())
    return float(v)


_UNSET = object()


def _is_vector_scale(scale):
    """One length scale per input dimension (torch tensor, NumPy array, list or tuple with more than one entry)?"""
    if torch.is_tensor(scale):
        return scale.numel() > 1
    if isinstance(scale, (list, tuple)):
        return True
    return np.ndim(scale) > 0 and np.size(scale) > 1


def _as_param(v):
    """Keep scalar tensors (they may carry an autograd graph: learnable hyper-parameters),
    turn everything else into a float."""
    if torch.is_tensor(v):
        if v.numel() != 1:
            raise ValueError("kernel hyper-parameters must be scalars")
        return v.reshape(())
    return float(v)


# ---------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------
class Kernel:
    """Base class.  A kernel that is a sum of stretched/scaled primitives exposes it as
    ``terms()`` -> list of ``(kind, variance, scale)``; other kernels override
    ``pairwise`` / ``elwise``."""

    stationary = False

    def terms(self):
        return None

    def tensor_terms(self):
        """Like ``terms()`` but variances / scales stay scalar tensors where the user gave
        tensors (so gradients can flow back to them)."""
        return self.terms()

    def num_outputs(self, x):
        return num_elements(x)

    def input_scaled_view(self):
        """``(k, scales)`` such that ``self(x, y) == k(x / scales, y / scales)`` with ``k`` a sum of primitives
        (``k.terms()`` is not None) and ``scales`` a vector of per-dimension length scales, or None for "inputs as they
        are"; None when the kernel has no such form.  The differentiable paths run ``k`` on the divided inputs and
        leave the division to torch, which carries the gradient to the length scales (and to ``x``)."""
        return (self, None) if self.terms() is not None else None

    # -- evaluation ------------------------------------------------------------
    def pairwise(self, x, y=None, *, lower=False, diag_add=0.0, diag_vec=None, cache=None, out=None):
        """``k(x, y)`` as a tensor (..., N, M); ``y is None``: symmetric case, where
        ``diag_add`` / ``diag_vec`` are added to the diagonal in the same pass.  ``out``: a
        (strided) matrix view to write into (blocks of a multi-output kernel matrix)."""
        t = self.terms()
        if t is None:
            raise NotImplementedError(f"pairwise evaluation is not implemented for {type(self).__name__}")
        x = uprank(x)
        y = None if y is None else uprank(y)
        if isinstance(x, MultiInput) or isinstance(y, MultiInput):
            raise ValueError(f"{type(self).__name__} is a single-output kernel; it cannot take multi-process inputs")
        return ops.get_backend().kmat(ops.KTerms(t), x, y, lower=lower, diag_add=diag_add, diag_vec=diag_vec, out=out)

    def elwise(self, x, y=None, *, cache=None):
        """``k(x_i, x_i)`` as a column (..., N, 1)."""
        if y is not None and y is not x:
            raise NotImplementedError("elwise is implemented for identical inputs")
        t = self.terms()
        if t is None:
            raise NotImplementedError(f"elwise evaluation is not implemented for {type(self).__name__}")
        return ops.get_backend().kdiag(ops.KTerms(t), uprank(x))[..., None]

    def __call__(self, x, y=None):
        """``k(x)`` / ``k(x, y)`` as a ``Dense`` matrix (lazily materialised for ``k(x)``)."""
        if y is None:
            return KernelDense(self, uprank(x), None)
        return Dense(self.pairwise(x, y))

    # -- algebra ---------------------------------------------------------------
    def __add__(self, other):
        if isinstance(other, (int, float)):
            if other == 0:
                return self
            other = other * OneKernel()
        if isinstance(other, ZeroKernel):
            return self
        if isinstance(self, ZeroKernel):
            return other
        if not isinstance(other, Kernel):
            return NotImplemented
        return Sum(self, other)

    __radd__ = __add__

    def __mul__(self, other):
        if isinstance(other, Kernel):
            raise NotImplementedError("products of kernels are outside the accelerated path")
        v = _as_param(other)
        if isinstance(self, ZeroKernel) or (not torch.is_tensor(v) and v == 0):
            return ZeroKernel()
        return Scaled(self, v)

    __rmul__ = __mul__

    def stretch(self, scale):
        """``k.stretch(l)``: ``k(x / l, y / l)``.  ``l``: a scalar, or one length scale per input dimension."""
        if _is_vector_scale(scale):
            return InputScaled(self, scale)
        return Stretched(self, _as_param(scale))

    def __reversed__(self):
        # sums of (stretched / scaled) primitives are symmetric, the zero kernel included
        return self if self.terms() is not None else Reversed(self)

    def __eq__(self, other):
        if isinstance(self, ZeroKernel) and isinstance(other, ZeroKernel):
            return True
        return self is other

    def __hash__(self):
        return id(self)


class _Primitive(Kernel):
    kind = None

    def terms(self):
        return [(self.kind, 1.0, 1.0)]

    def __repr__(self):
        return f"{type(self).__name__}()"


class EQ(_Primitive):
    """Exponentiated quadratic ``exp(-r^2 / 2)``."""
    kind = "eq"
    stationary = True


class Matern12(_Primitive):
    """Exponential kernel ``exp(-r)``."""
    kind = "matern12"
    stationary = True


Exp = Matern12


class Matern32(_Primitive):
    kind = "matern32"
    stationary = True


class Matern52(_Primitive):
    kind = "matern52"
    stationary = True


class Linear(_Primitive):
    """``<x, y>``."""
    kind = "linear"


class OneKernel(_Primitive):
    kind = "const"
    stationary = True


class ZeroKernel(Kernel):
    stationary = True

    def terms(self):
        return []

    def pairwise(self, x, y=None, *, out=None, diag_add=0.0, diag_vec=None, **kw):
        x = uprank(x)
        sym = y is None
        y = x if sym else uprank(y)
        shape = tuple(x.shape[:-1]) + (y.shape[-2],)
        if out is not None:               # a caller's (strided) view: the zeros go THERE (blocks of one buffer, rows under a matrix)
            if tuple(out.shape) != shape:
                raise ValueError(f"out has shape {tuple(out.shape)}, the kernel matrix {shape}")
            out.zero_()
        else:
            out = torch.zeros(shape, dtype=x.dtype, device=x.device)
        return _add_diag(out, diag_add, diag_vec) if sym else out      # (noise / jitter of `k(x) + noise`: the diagonal is all there is)

    def elwise(self, x, y=None, **kw):
        x = uprank(x)
        return torch.zeros(x.shape[:-1] + (1,), dtype=x.dtype, device=x.device)

    def __repr__(self):
        return "0"


class Scaled(Kernel):
    def __init__(self, k, v):
        self.k, self.v = k, v
        self.stationary = k.stationary

    def num_outputs(self, x):
        return self.k.num_outputs(x)

    def terms(self):
        t = self.k.terms()
        return None if t is None else [(kind, var * _as_float(self.v), s) for kind, var, s in t]

    def tensor_terms(self):
        t = self.k.tensor_terms()
        return None if t is None else [(kind, var * self.v, s) for kind, var, s in t]

    def input_scaled_view(self):
        v = self.k.input_scaled_view()
        return None if v is None else (Scaled(v[0], self.v), v[1])

    def pairwise(self, x, y=None, **kw):
        if self.terms() is not None:
            return super().pairwise(x, y, **kw)
        view = self.input_scaled_view()
        if view is not None:                       # v * k0(x / l): still ONE fused launch, on the divided inputs
            return InputScaled(*view).pairwise(x, y, **kw)
        da, dv = kw.pop("diag_add", 0.0), kw.pop("diag_vec", None)
        out = _as_float(self.v) * self.k.pairwise(x, y, **kw)
        return _add_diag(out, da, dv) if y is None else out

    def elwise(self, x, y=None, **kw):
        if self.terms() is not None:
            return super().elwise(x, y, **kw)
        view = self.input_scaled_view()
        if view is not None:
            return InputScaled(*view).elwise(x, y, **kw)
        return _as_float(self.v) * self.k.elwise(x, y, **kw)

    def __repr__(self):
        return f"{self.v} * {self.k!r}"


class Stretched(Kernel):
    def __init__(self, k, scale):
        if k.terms() is None:
            raise NotImplementedError("stretch is implemented for sums of primitive kernels")
        self.k, self.scale = k, scale
        self.stationary = k.stationary

    def terms(self):
        return [(kind, var, s * _as_float(self.scale)) for kind, var, s in self.k.terms()]

    def tensor_terms(self):
        return [(kind, var, s * self.scale) for kind, var, s in self.k.tensor_terms()]

    def __repr__(self):
        return f"({self.k!r} > {self.scale})"


class InputScaled(Kernel):
    """``k.stretch(l)`` with one length scale per input dimension (mlkernels' ``Stretched`` with a vector:
    ``k(x / l, y / l)``).  The inputs are divided once -- O(N D) -- and the inner kernel (a sum of primitives) runs
    its fused kernel-matrix launch on them, lower-triangle-only / in-place-factorisable like any other; sums of
    kernels with DIFFERENT per-dimension scales are evaluated term group by term group (``Sum.pairwise``)."""

    def __init__(self, k, scales):
        if k.terms() is None:
            raise NotImplementedError("stretch is implemented for sums of primitive kernels")
        # (a tensor is kept AS IT IS -- the caller's object: in-place updates and autograd leaves must stay visible; under a
        # `torch.set_default_device(...)` mode `torch.as_tensor(tensor)` would silently copy it to that device)
        if not torch.is_tensor(scales):
            scales = torch.as_tensor(scales)
        if scales.dim() != 1:
            raise ValueError("per-dimension length scales are a vector (one entry per input dimension)")
        if not bool((scales > 0).all()):
            raise ValueError("length scales must be positive")
        self.k, self.scales = k, scales
        self.stationary = k.stationary

    def _scaled(self, x):
        x = uprank(x)
        if isinstance(x, MultiInput):
            raise ValueError("InputScaled is a single-output kernel; it cannot take multi-process inputs")
        if x.shape[-1] != self.scales.numel():
            raise ValueError(f"inputs have {x.shape[-1]} dimensions, the kernel has {self.scales.numel()} length scales")
        # (values only: the differentiable log-density / bound take the kernel apart through `input_scaled_view`
        # and divide the inputs themselves; everything that is not covered there refuses loudly at its own entry)
        return x.detach() / self.scales.detach().to(dtype=x.dtype, device=x.device)

    def input_scaled_view(self):
        return (self.k, self.scales)

    def num_outputs(self, x):
        return self.k.num_outputs(x)

    def pairwise(self, x, y=None, **kw):
        return self.k.pairwise(self._scaled(x), None if y is None else self._scaled(y), **kw)

    def elwise(self, x, y=None, **kw):
        if y is not None and y is not x:
            raise NotImplementedError("elwise is implemented for identical inputs")
        return self.k.elwise(self._scaled(x), **kw)

    def stretch(self, scale):
        if _is_vector_scale(scale):
            return InputScaled(self.k, self.scales * (scale if torch.is_tensor(scale) else torch.as_tensor(scale)).to(self.scales))
        return InputScaled(self.k, self.scales * _as_float(scale))

    def __reversed__(self):
        return self                      # k(x / l, y / l) of a symmetric k is symmetric

    def __repr__(self):
        return f"({self.k!r} > {self.scales.tolist()})"


def _merge_terms(terms):
    """Add up the variances of terms with the same primitive and the same length scale (``p + p``
    has the kernel ``4 k``, not four terms; the fused kernels take at most 8 distinct terms)."""
    out = []
    for kind, var, scale in terms:
        for i, (k2, v2, s2) in enumerate(out):
            same = (scale is s2) if (torch.is_tensor(scale) or torch.is_tensor(s2)) else (scale == s2)
            if k2 == kind and same:
                out[i] = (k2, v2 + var, s2)
                break
        else:
            out.append((kind, var, scale))
    return out


def _scale_stamp(t):
    """Identity + version of a length-scale vector (``None`` scale: a constant stamp; a tensor nothing vouches for: ``None``)."""
    if t is None:
        return (0, 0)
    if not torch.is_tensor(t) or t.is_inference() or not t.is_cuda:
        return None
    return (id(t), t._version)


class Sum(Kernel):
    def __init__(self, a, b):
        self.a, self.b = a, b
        self.stationary = a.stationary and b.stationary
        self._view = _UNSET
        self._view_stamp = None

    def num_outputs(self, x):
        return self.a.num_outputs(x)

    def terms(self):
        ta, tb = self.a.terms(), self.b.terms()
        if ta is None or tb is None:
            return None
        return _merge_terms(ta + tb)

    def tensor_terms(self):
        ta, tb = self.a.tensor_terms(), self.b.tensor_terms()
        if ta is None or tb is None:
            return None
        return _merge_terms(ta + tb)

    def input_scaled_view(self):
        # asked several times per FDD / log-density: the comparison of the two length-scale vectors (a device-to-host
        # synchronisation when they live on the GPU) is made once per kernel object
        # -- and again whenever one of them has been written in place since (the answer of `torch.equal` must not outlive
        # the values it compared; tensors without a version counter are compared every time)
        va, vb = self.a.input_scaled_view(), self.b.input_scaled_view()
        stamp = tuple(_scale_stamp(v[1]) for v in (va, vb) if v is not None)
        if self._view is _UNSET or None in stamp or stamp != self._view_stamp:
            self._view = self._input_scaled_view()
            self._view_stamp = stamp
        return self._view

    def _input_scaled_view(self):
        va, vb = self.a.input_scaled_view(), self.b.input_scaled_view()
        if va is None or vb is None:
            return None
        sa, sb = va[1], vb[1]
        same = (sa is sb) or (sa is not None and sb is not None and sa.shape == sb.shape and not sa.requires_grad
                              and not sb.requires_grad
                              and (sa.data_ptr() == sb.data_ptr() and sa.device == sb.device and sa.dtype == sb.dtype
                                   or bool(torch.equal(sa, sb.to(sa)))))
        return (Sum(va[0], vb[0]), sa) if same else None      # different length-scale vectors: no common division

    def pairwise(self, x, y=None, **kw):
        if self.terms() is not None:
            return super().pairwise(x, y, **kw)
        view = self.input_scaled_view()
        if view is not None:                       # both summands divide the inputs by the same length scales
            return InputScaled(*view).pairwise(x, y, **kw)
        da, dv = kw.pop("diag_add", 0.0), kw.pop("diag_vec", None)
        kw.pop("lower", None)
        out = self.a.pairwise(x, y, **kw) + self.b.pairwise(x, y, **kw)
        return _add_diag(out, da, dv) if y is None else out

    def elwise(self, x, y=None, **kw):
        if self.terms() is not None:
            return super().elwise(x, y, **kw)
        view = self.input_scaled_view()
        if view is not None:
            return InputScaled(*view).elwise(x, y, **kw)
        return self.a.elwise(x, y, **kw) + self.b.elwise(x, y, **kw)

    def __repr__(self):
        return f"{self.a!r} + {self.b!r}"


class Reversed(Kernel):
    """``reversed(k)(x, y) = k(y, x)^T`` (used for cross-kernels, measure.py:111)."""

    def __init__(self, k):
        self.k = k
        self.stationary = k.stationary

    def num_outputs(self, x):
        return self.k.num_outputs(x)

    def terms(self):
        return self.k.terms()   # sums of primitives are symmetric

    def tensor_terms(self):
        return self.k.tensor_terms()

    def pairwise(self, x, y=None, **kw):
        if self.terms() is not None or y is None:
            return self.k.pairwise(x, y, **kw)
        return self.k.pairwise(y, x, **kw).transpose(-1, -2)

    def elwise(self, x, y=None, **kw):
        return self.k.elwise(x, y, **kw)


# ---------------------------------------------------------------------------
# Cartesian products of processes (stheno/mo/kernel.py, stheno/mo/mean.py, measure.py:404-423)
# ---------------------------------------------------------------------------
def _eval_into(k, x, y, view, lower=False):
    """Write ``k(x, y)`` (``y is None``: symmetric) into the matrix view ``view``: a sum of
    primitives goes straight into the block (one fused launch, no temporary)."""
    if isinstance(k, ZeroKernel):
        view.zero_()
    elif k.terms() is not None and not isinstance(y, MultiInput):
        k.pairwise(x, y, lower=lower, out=view)
    else:
        view.copy_(k.pairwise(x, y))


class MultiOutputKernel(Kernel):
    """Kernel of the Cartesian product of the processes ``ps`` of a measure: block ``(i, j)``
    of ``k(X, Y)`` is ``measure.kernels[p_i, p_j](x_i, y_j)`` (``stheno/mo/kernel.py:39-76``,
    ``stheno/mo/input.py:7-9``).  A plain input stands for "every process at these inputs".
    The block matrix is assembled in ONE buffer (each block is written in place through its
    leading dimension); in the symmetric ``lower`` mode the blocks above the diagonal are skipped,
    so the buffer can be Cholesky-factorised in place like a single-process kernel matrix.

    The measure is referenced weakly and the processes by id (the kernel lives in the measure's
    own store); a ``MultiInput`` keeps its processes alive."""

    def __init__(self, measure, ps):
        import weakref

        self._measure = weakref.ref(measure)
        self.pids = tuple(id(p) for p in ps)

    @property
    def kernels(self):
        return self._measure().kernels

    def _split(self, x):
        if hasattr(x, "p") and hasattr(x, "_xr"):          # an FDD: that process at those inputs
            xr = x._xr
            return [(id(q), xi) for q, xi in xr.parts] if isinstance(xr, MultiInput) else [(id(x.p), xr)]
        if isinstance(x, (tuple, list)):                   # several FDDs (mo/input.py:7-9)
            return [part for e in x for part in self._split(e)]
        x = uprank(x)
        if isinstance(x, MultiInput):
            return [(id(p), xi) for p, xi in x.parts]
        return [(pid, x) for pid in self.pids]

    def num_outputs(self, x):
        kernels = self.kernels
        return sum(kernels[pi].num_outputs(xi) for pi, xi in self._split(x))

    def pairwise(self, x, y=None, *, lower=False, diag_add=0.0, diag_vec=None, cache=None, out=None):
        kernels = self.kernels
        sym = y is None
        X = self._split(x)
        Y = X if sym else self._split(y)
        # a part's process may itself be a product process (then its plain input expands): sizes come
        # from the processes' own kernels (the reference marks such kernels "ADK": stheno/mo/adk.py)
        sx = [kernels[pi].num_outputs(xi) for pi, xi in X]
        sy = sx if sym else [kernels[pj].num_outputs(yj) for pj, yj in Y]
        nx, ny = sum(sx), sum(sy)
        if out is None:
            out = ops.alloc_matrix(tuple(X[0][1].shape[:-2]), nx, ny, X[0][1].dtype, X[0][1].device)
        r0 = 0
        for i, (pi, xi) in enumerate(X):
            r1, c0 = r0 + sx[i], 0
            for j, (pj, yj) in enumerate(Y):
                c1 = c0 + sy[j]
                if not (sym and lower and j > i) and r1 > r0 and c1 > c0:
                    if sym and i == j:
                        _eval_into(kernels[pi], xi, None, out[..., r0:r1, c0:c1], lower=lower)
                    else:
                        _eval_into(kernels[pi, pj], xi, yj, out[..., r0:r1, c0:c1])
                c0 = c1
            r0 = r1
        return _add_diag(out, diag_add, diag_vec) if sym else out

    def elwise(self, x, y=None, *, cache=None):
        if y is not None and y is not x:
            raise NotImplementedError("elwise is implemented for identical inputs")
        kernels = self.kernels
        return torch.cat([kernels[pi].elwise(xi) for pi, xi in self._split(x)], dim=-2)

    def __repr__(self):
        return "MultiOutputKernel(" + ", ".join(repr(self.kernels[pid]) for pid in self.pids) + ")"


class _CrossKernel(Kernel):
    """``k(p_cross, p_j)``: rows follow the multi-input of the product process, columns the
    inputs of process ``j`` (the left rule of ``Measure.cross``, ``measure.py:419-422``)."""

    def __init__(self, mok, j):
        self.mok, self.j = mok, j

    def num_outputs(self, x):
        return self.mok.num_outputs(x)

    def pairwise(self, x, y=None, *, cache=None, **kw):
        if y is None:
            raise ValueError("a cross-kernel between a product process and another process needs both inputs")
        kernels = self.mok.kernels
        X = self.mok._split(x)
        y = uprank(y)
        sx = [kernels[pi].num_outputs(xi) for pi, xi in X]
        nx, ny = sum(sx), kernels[self.j].num_outputs(y)
        out = ops.alloc_matrix(tuple(X[0][1].shape[:-2]), nx, ny, X[0][1].dtype, X[0][1].device)
        r0 = 0
        for (pi, xi), n_i in zip(X, sx):
            r1 = r0 + n_i
            if r1 > r0 and ny > 0:
                _eval_into(kernels[pi, self.j], xi, y, out[..., r0:r1, :])
            r0 = r1
        return out

    def elwise(self, x, y=None, **kw):
        raise ValueError('Unclear combination of arguments given to "elwise".')    # mo/kernel.py:63-70


def _add_diag(out, diag_add, diag_vec):
    if diag_add or diag_vec is not None:
        ops.get_backend().add_diag_(out, diag_add, diag_vec)
    return out


# ---------------------------------------------------------------------------
# means
# ---------------------------------------------------------------------------
class Mean:
    def __call__(self, x, cache=None):
        raise NotImplementedError

    def __add__(self, other):
        if isinstance(other, (int, float)) and other == 0:
            return self
        return SumMean(self, other if isinstance(other, Mean) else _wrap_mean(other))

    __radd__ = __add__

    def __mul__(self, other):
        return ScaledMean(self, _as_float(other))

    __rmul__ = __mul__


class ZeroMean(Mean):
    def __call__(self, x, cache=None):
        x = uprank(x)
        return torch.zeros(x.shape[:-1] + (1,), dtype=x.dtype, device=x.device)

    def __repr__(self):
        return "0"


class OneMean(Mean):
    def __call__(self, x, cache=None):
        x = uprank(x)
        return torch.ones(x.shape[:-1] + (1,), dtype=x.dtype, device=x.device)

    def __repr__(self):
        return "1"


class FunctionMean(Mean):
    """Mean given by a Python function of the (upranked) inputs, e.g. ``lambda x: x ** 2``."""

    def __init__(self, f):
        self.f = f

    def __call__(self, x, cache=None):
        return uprank(self.f(uprank(x)))

    def __repr__(self):
        return getattr(self.f, "__name__", "f")


class ScaledMean(Mean):
    def __init__(self, m, v):
        self.m, self.v = m, v

    def __call__(self, x, cache=None):
        return self.v * self.m(x, cache)

    def __repr__(self):
        return f"{self.v} * {self.m!r}"


class SumMean(Mean):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def __call__(self, x, cache=None):
        return self.a(x, cache) + self.b(x, cache)

    def __repr__(self):
        return f"{self.a!r} + {self.b!r}"


def _wrap_mean(m):
    if isinstance(m, Mean):
        return m
    if callable(m):
        return FunctionMean(m)
    v = _as_float(m)
    if v == 0:
        return ZeroMean()
    return OneMean() if v == 1 else ScaledMean(OneMean(), v)


class MultiOutputMean(Mean):
    """Mean of the Cartesian product: the per-process means stacked (``stheno/mo/mean.py``)."""

    def __init__(self, measure, ps):
        import weakref

        self._measure = weakref.ref(measure)
        self.pids = tuple(id(p) for p in ps)

    def __call__(self, x, cache=None):
        means = self._measure().means
        x = uprank(x)
        parts = [(id(p), xi) for p, xi in x.parts] if isinstance(x, MultiInput) else [(pid, x) for pid in self.pids]
        return torch.cat([_call_mean(means[pi], xi, cache) for pi, xi in parts], dim=-2)


# ---------------------------------------------------------------------------
# posterior objects
# ---------------------------------------------------------------------------
def _cross(cache, k, z, x):
    """Memoised ``k(z, x)`` tensor within one ``mean_var(_diag)`` evaluation: the
    reference evaluates the data/inducing cross-kernel once per prediction
    (pinned by ``tests/model/test_model.py:335-365``).  Entries are keyed by object identity and
    keep their key objects alive, so an id cannot be recycled while the cache lives."""
    if cache is None:
        return k.pairwise(z, x)
    key = ("kzx", id(k), id(z), id(x))
    if key not in cache:
        cache[key] = (k.pairwise(z, x), k, z, x)
    return cache[key][0]


class WhitenedT:
    """``k(x, z) L^{-T}`` (ns, n): the TRANSPOSE of the whitened cross-covariance ``L^{-1} k(z, x)`` -- what the factorisation with
    rows under the matrix leaves behind (``KernelDense.chol_with_rows``).  The consumers below take either form."""

    def __init__(self, zt):
        self.zt = zt

    def plain(self):
        return self.zt.transpose(-1, -2).contiguous()


def _whiten(cache, K_z, k, z, x, own_cross=False, rhs=None):
    """``L^{-1} k(z, x)`` (memoised), ``L = chol(K_z)``.  ``own_cross``: nobody else will ask for ``k(z, x)`` itself (dense
    conditioning: ``K_z`` is the only matrix it is ever solved against) -- the cross matrix is then not kept and the blocked solve
    takes it as its workspace instead of a copy (N x N* elements: 268 MB and 0.1 ms at cfg2).
    ``rhs = (r, took)``: if THIS call is what factorises ``K_z`` (rows under the matrix), the residual ``r`` (n, 1) rides along as a
    right-hand side (``matrix.config.posterior_rows_rhs``) and ``took(w)`` is called with ``w = L^{-1} r``."""
    key = ("v", id(K_z), id(k), id(z), id(x))
    if cache is not None and key in cache:
        return cache[key][0]
    # Nothing factorised yet (conditioning, then prediction, no log-density in between): the cross-covariance rides in the
    # factorisation as rows under the kernel matrix and comes out whitened, transposed -- no separate many-column solve.
    # (a non-empty term list: the zero kernel -- the cross-kernel of a process independent of the observed one -- has `terms() == []`,
    #  nothing to whiten and no business in the factorisation)
    if (own_cross and hasattr(K_z, "can_factor_with_rows") and k.terms() and not isinstance(k, ZeroKernel) and x.dim() == 2 and z.dim() == 2 and not x.requires_grad
            and ("kzx", id(k), id(z), id(x)) not in (cache or {}) and z is getattr(K_z, "x", None) and K_z.can_factor_with_rows(x.shape[-2])):
        from .matrix import config as _mconfig
        if rhs is not None and _mconfig.posterior_rows_rhs:
            got = K_z.chol_with_rows(k, x, rhs=rhs[0])
            if len(got) == 3:
                rhs[1](got[2])
            zt = got[1]
        else:
            _, zt = K_z.chol_with_rows(k, x)
        v = WhitenedT(zt)
        if cache is not None:
            cache[key] = (v, K_z, k, z, x)
        return v
    if isinstance(k, ZeroKernel):         # an independent process: L^{-1} 0 = 0, nothing to solve
        v = k.pairwise(z, x)
        if cache is not None:
            cache[key] = (v, K_z, k, z, x)
        return v
    chol = K_z.chol()
    if own_cross and hasattr(chol, "solve_") and ("kzx", id(k), id(z), id(x)) not in (cache or {}):
        kzx = k.pairwise(z, x)
        same_batch = kzx.dim() == chol.l.dim() and tuple(kzx.shape[:-2]) == tuple(chol.l.shape[:-2])
        v = chol.solve_(kzx) if (same_batch and kzx.is_contiguous() and not kzx.requires_grad and kzx.shape[-2] == chol.n) else chol.solve(kzx)
    else:
        v = chol.solve(_cross(cache, k, z, x))
    if cache is not None:
        cache[key] = (v, K_z, k, z, x)
    return v


class PosteriorKernel(Kernel):
    """``k_ij(x, y) - k_zi(z, x)^T K_z^{-1} k_zj(z, y)`` (mlkernels.PosteriorKernel;
    constructed at ``observations.py:148-154,256-261``)."""

    def __init__(self, k_ij, k_zi, k_zj, z, K_z, own_cross=False):
        self.k_ij, self.k_zi, self.k_zj, self.z, self.K_z = k_ij, k_zi, k_zj, z, K_z
        self.own_cross = own_cross       # see _whiten

    def num_outputs(self, x):
        return self.k_ij.num_outputs(x)

    def pairwise(self, x, y=None, *, lower=False, diag_add=0.0, diag_vec=None, cache=None):
        x = uprank(x)
        sym = y is None
        y = x if sym else uprank(y)
        out = self.k_ij.pairwise(x, None if sym else y, cache=cache) if _accepts_cache(self.k_ij) else \
            self.k_ij.pairwise(x, None if sym else y)
        vx = _whiten(cache, self.K_z, self.k_zi, self.z, x, self.own_cross)
        vy = vx if (sym and self.k_zi is self.k_zj) else _whiten(cache, self.K_z, self.k_zj, self.z, y, self.own_cross)
        # out -= vx^T vy   (operands stored (K, M) / (K, N): row index contiguous)
        if out.stride(-1) != 1 and out.shape[-1] > 1:
            out = out.contiguous()          # e.g. the transposed view a reversed cross-kernel hands back
        if isinstance(vx, WhitenedT) and isinstance(vy, WhitenedT):      # both transposed: the k-contiguous form of the same product
            ops.get_backend().gemm(vx.zt, vy.zt, a_kmajor=True, b_kmajor=True, alpha=-1.0, beta=1.0, out=out)
        else:
            vx, vy = (v.plain() if isinstance(v, WhitenedT) else v for v in (vx, vy))
            ops.get_backend().gemm(vx, vy, a_kmajor=False, b_kmajor=False, alpha=-1.0, beta=1.0, out=out)
        return _add_diag(out, diag_add, diag_vec) if sym else out

    def elwise(self, x, y=None, *, cache=None):
        x = uprank(x)
        out = self.k_ij.elwise(x)
        vx = _whiten(cache, self.K_z, self.k_zi, self.z, x, self.own_cross)
        if self.k_zi is self.k_zj:
            if isinstance(vx, WhitenedT):
                # (the posterior mean of the same evaluation has read these rows already and left their sums of squares: one pass)
                hit = cache.pop(("rowss", id(vx)), None) if cache is not None else None
                if hit is not None and hit[1] is vx:
                    ss = hit[0]
                else:
                    _, ss = ops.get_backend().rowreduce(vx.zt, want_dot=False, want_ss=True)
            else:
                _, ss = ops.get_backend().colreduce(vx, want_ss=True)
            return out - ss[..., None]
        vy = _whiten(cache, self.K_z, self.k_zj, self.z, x, self.own_cross)
        vx, vy = (v.plain() if isinstance(v, WhitenedT) else v for v in (vx, vy))
        return out - (vx * vy).sum(-2)[..., None]


class SubspaceKernel(Kernel):
    """``k_zi(z, x)^T A^{-1} k_zj(z, y)`` (mlkernels.SubspaceKernel; ``observations.py:262-267``)."""

    def __init__(self, k_zi, k_zj, z, A):
        self.k_zi, self.k_zj, self.z, self.A = k_zi, k_zj, z, A

    def pairwise(self, x, y=None, *, lower=False, diag_add=0.0, diag_vec=None, cache=None):
        x = uprank(x)
        sym = y is None
        y = x if sym else uprank(y)
        vx = _whiten(cache, self.A, self.k_zi, self.z, x)
        vy = vx if (sym and self.k_zi is self.k_zj) else _whiten(cache, self.A, self.k_zj, self.z, y)
        out = ops.get_backend().gemm(vx, vy, a_kmajor=False, b_kmajor=False)
        return _add_diag(out, diag_add, diag_vec) if sym else out

    def elwise(self, x, y=None, *, cache=None):
        x = uprank(x)
        vx = _whiten(cache, self.A, self.k_zi, self.z, x)
        if self.k_zi is self.k_zj:
            _, ss = ops.get_backend().colreduce(vx, want_ss=True)
            return ss[..., None]
        vy = _whiten(cache, self.A, self.k_zj, self.z, x)
        return (vx * vy).sum(-2)[..., None]


def _accepts_cache(k):
    return isinstance(k, (PosteriorKernel, SubspaceKernel, Sum, Scaled, Reversed))


class PosteriorMean(Mean):
    """``m_i(x) + k_zi(z, x)^T K_z^{-1} (y - m_z(z))`` (mlkernels.PosteriorMean;
    ``observations.py:161-168,270-277``), evaluated as ``(L^{-1} k_zx)^T (L^{-1} (y - m_z(z)))``."""

    def __init__(self, m_i, m_z, k_zi, z, K_z, y, own_cross=False):
        self.m_i, self.m_z, self.k_zi, self.z, self.K_z, self.y = m_i, m_z, k_zi, z, K_z, y
        self.own_cross = own_cross       # see _whiten
        self._w = None
        self._r = None

    def _residual(self):
        if self._r is None:
            y = uprank(self.y)
            # (a zero mean: the residual IS the data -- nobody below writes into it)
            self._r = (y, y if (isinstance(self.m_z, ZeroMean) and torch.is_tensor(y) and not y.requires_grad) else y - self.m_z(self.z))
        return self._r

    def _whitened_residual(self):
        if self._w is None:
            y, r = self._residual()
            chol = self.K_z.chol()
            if isinstance(self.m_z, ZeroMean) and hasattr(chol, "solve_residual"):
                self._w = chol.solve_residual(r, y)     # shared with the log-density of the same observations
            else:
                self._w = chol.solve(r)                 # (..., N, 1)
        return self._w

    def _took(self, w):
        """``w = L^{-1} (y - m_z(z))`` came out of the factorisation (a right-hand side under the matrix): kept here and, for a zero
        mean, filed with the factor under ``y`` -- the log-density of the same observations finds it (``Chol.solve_residual``)."""
        self._w = w
        if isinstance(self.m_z, ZeroMean):
            chol = self.K_z.chol()
            if hasattr(chol, "remember_residual"):
                chol.remember_residual(self._residual()[0], w)

    def __call__(self, x, cache=None):
        x = uprank(x)
        rhs = None
        if self._w is None and self.own_cross and hasattr(self.K_z, "can_factor_with_rows") and x.dim() == 2:
            y, r = self._residual()
            if r.dim() == 2 and r.shape[-1] == 1 and not r.requires_grad:
                rhs = (r, self._took)
        v = _whiten(cache, self.K_z, self.k_zi, self.z, x, self.own_cross, rhs)       # (first: it may be what factorises K_z)
        w = self._whitened_residual()
        if isinstance(v, WhitenedT):
            if w.shape[-1] == 1:
                # mean AND the rows' sums of squares (what the marginal variance of the same evaluation subtracts, PosteriorKernel.elwise)
                # from ONE pass over the whitened rows: they are N* x N elements, read from HBM either way
                dot, ss = ops.get_backend().rowreduce(v.zt, w, want_dot=True, want_ss=cache is not None)
                if cache is not None and ss is not None:
                    cache[("rowss", id(v))] = (ss, v)
                return self.m_i(x) + dot[..., None]
            v = v.plain()
        dot, _ = ops.get_backend().colreduce(v, w, want_dot=True, want_ss=False)
        return self.m_i(x) + dot[..., None]


# ---------------------------------------------------------------------------
# mean_var / mean_var_diag  (fdd.py:68-74)
# ---------------------------------------------------------------------------
def _call_mean(mean, x, cache):
    return mean(x, cache) if isinstance(mean, Mean) else mean(x)


def mean_var(mean, kernel, x):
    """Mean and variance sharing the cross-kernel evaluations."""
    from .matrix import deferred_checks

    cache = {}
    x = uprank(x)
    # (a posterior evaluated before anything factorised its observations factorises here: the `info` word is read when the
    # reductions / products that depend on the factor are queued behind it, not in front of them -- as `Normal.logpdf` does)
    with deferred_checks():
        m = _call_mean(mean, x, cache)
        if _accepts_cache(kernel):
            var = Dense(kernel.pairwise(x, None, cache=cache))
        else:
            var = KernelDense(kernel, x, None)
    return m, var


def mean_var_diag(mean, kernel, x):
    """Mean and marginal variances without forming the covariance
    (``tests/model/test_gp.py:201-211``)."""
    from .matrix import deferred_checks

    cache = {}
    x = uprank(x)
    with deferred_checks():          # (see mean_var)
        m = _call_mean(mean, x, cache)
        vd = kernel.elwise(x, cache=cache) if _accepts_cache(kernel) else kernel.elwise(x)
    return m, vd
