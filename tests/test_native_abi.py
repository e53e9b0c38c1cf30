"""The C ABI boundary: libgpk.so builds, loads, and exports exactly what include/gpk.h
declares (no compute calls: this runs on the CPU box)."""
import ctypes
import os
import re

import pytest

from stheno_amd import _native

from .conftest import ROOT


def _declared():
    with open(os.path.join(ROOT, "include", "gpk.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpk_[a-z0-9_]+)\s*\(", text)))


def _ensure_built():
    if not os.path.exists(_native.lib_path()):
        import __graft_entry__ as g

        g.build()


def test_header_and_binding_agree():
    assert _declared() == sorted(_native.SIGNATURES)


def test_library_loads_and_exports_every_declared_symbol():
    _ensure_built()
    lib = ctypes.CDLL(_native.lib_path())
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/gpk.h but not exported by libgpk.so"
    # ... and nothing else: every exported gpk_* symbol is declared (no undocumented back doors)
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _native.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in out.splitlines() if re.match(r"^[0-9a-f]+ T gpk_[a-z0-9_]+$", line.strip())})
    assert exported == _declared()
    assert _native.load().gpk_version() >= 100
    assert _native.load().gpk_dinv_elems(300) == 3 * 128 * 128


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_LIB", None)
    monkeypatch.setattr(_native, "lib_path", lambda: str(tmp_path / "libgpk.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.load()


def test_cpu_tensors_are_rejected_by_the_hip_backend():
    import torch

    from stheno_amd import ops

    _ensure_built()
    be = ops.HipBackend()
    x = torch.randn(5, 2, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="HIP device only"):
        be.kmat(ops.KTerms([("eq", 1.0, 1.0)]), x)
