"""Round 6 evidence on the MI355X (all ``-m gpu``, through ``libgpk.so``)."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O
from stheno_amd import _native, matrix

from .conftest import ROOT

pytestmark = pytest.mark.gpu

OUT = os.path.join(ROOT, "gpurun_out", "r06")


def _note(name, rec):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "achieved.json")
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[name] = rec
    with open(path, "w") as f:
        json.dump(data, f, indent=1)
    print("ACHIEVED", name, json.dumps(rec))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_measured_mfma_peak_is_a_plausible_ceiling(hip_backend, dtype):
    """``gpk_mfma_peak`` (the peak re-derived on the box): a register-resident MFMA stream on every SIMD cannot beat the
    nominal peak, and on a healthy MI355X it reaches a good part of it; the clock it reports is a real shader clock."""
    lib = _native.load()
    nominal = {"f64": 78.6, "f32": 157.3}[dtype]
    best = 0.0
    for waves in (1, 2):
        tf, ms, mhz, eff = (ctypes.c_double() for _ in range(4))
        st_ = lib.gpk_mfma_peak(0 if dtype == "f32" else 1, 50.0, waves, ctypes.byref(tf), ctypes.byref(ms), ctypes.byref(mhz), ctypes.byref(eff),
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert st_ == 0
        assert ms.value >= 50.0
        assert 0.3 * nominal < tf.value <= 1.02 * nominal, (dtype, waves, tf.value)
        assert 500.0 < mhz.value <= 2500.0, mhz.value
        assert 0.3 < eff.value <= 1.02, eff.value
        best = max(best, tf.value)
        _note(f"mfma_peak_{dtype}_{waves}w", {"tflops": tf.value, "ms": ms.value, "shader_clock_mhz": mhz.value, "issue_eff": eff.value})
    assert lib.gpk_mfma_peak(1, 1.0, 3, ctypes.byref(tf), None, None, None, None) == -3
    assert lib.gpk_mfma_peak(7, 1.0, 2, ctypes.byref(tf), None, None, None, None) == -1


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n,S,pad", [(300, 5, 0), (512, 25, 1), (1000, 3, 4), (64, 1, 0)])
def test_sum_of_split_k_parts_over_the_lower_triangle(hip_backend, dtype, n, S, pad):
    """``gpk_sum_lower``: the partial products of the pseudo-point path's split-K SYRK (observations.py:322) added up natively, lower
    triangle only, into a strided view (the statistics buffer has one column more than the matrix); nothing above the diagonal of a
    row's last vector is written, what lies above the diagonal in the parts is never read as part of the result."""
    g = torch.Generator(device="cpu").manual_seed(n + S)
    parts = torch.randn(S, n, n, generator=g, dtype=torch.float64).to(dtype).cuda()
    ref = torch.tril(parts.to(torch.float64).sum(0))
    parts_nan = parts.clone()
    iu = torch.triu_indices(n, n, 128 * ((0 + 127) // 128) + 128)      # strictly above the diagonal TILES: never read
    if iu.numel():
        parts_nan[:, iu[0], iu[1]] = float("nan")
    buf = torch.full((n, n + pad), 7.0, dtype=dtype, device="cuda")
    out = buf[:, :n]
    hip_backend.sum_lower(parts_nan, out)
    got = torch.tril(out).to(torch.float64)
    tol = 1e-13 if dtype == torch.float64 else 1e-5
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= tol * float(ref.abs().max())
    if pad:
        assert float(buf[:, n:].min()) == 7.0 and float(buf[:, n:].max()) == 7.0
    # far above the diagonal nothing was touched
    if n > 300:
        assert float(out[0, 300:].min()) == 7.0


def test_batched_factorisation_on_a_cu_masked_stream_keeps_to_the_lockstep_launches(hip_backend):
    """The mixed-phase batched steps rest on block L of a launch running on XCD L % 8 (all tasks of a matrix on one XCD: no fences).  A
    stream with a CU mask may not see every XCD: ``potrf_plain`` asks the runtime for the stream's mask and keeps such streams on the
    lockstep launches.  Here: 72 fp32 matrices of order 1024 on a stream confined to the first 32 CU-mask bits, against the default
    stream -- the same factors bit for bit (both paths run the same arithmetic in the same order), ``info`` clean."""
    hip = ctypes.CDLL("libamdhip64.so")
    stream = ctypes.c_void_p()
    mask = (ctypes.c_uint32 * 8)(0xffffffff, 0, 0, 0, 0, 0, 0, 0)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), 8, mask)
    if rc != 0:
        pytest.skip("the runtime refuses CU-masked streams here")
    try:
        g = torch.Generator(device="cpu").manual_seed(3)
        x = torch.randn(72, 1024, 3, generator=g, dtype=torch.float32).cuda()
        k = hip_backend.kmat(st.ops.KTerms([("eq", 1.0, 1.0)]), x, None, lower=True, diag_add=0.1)
        a0, a1 = k.clone(), k.clone()
        lib = _native.load()

        def mixed_launches():
            ms, nl, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
            lib.gpk_prof_stop(160, ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(fl))
            return nl.value

        lib.gpk_prof_start()
        dinv0, info0 = hip_backend.potrf_(a0)
        torch.cuda.synchronize()
        assert mixed_launches() == 1024 // 128 - 1          # the default stream: one mixed-phase launch per step below the last block
        ext = torch.cuda.ExternalStream(stream.value)
        lib.gpk_prof_start()
        with torch.cuda.stream(ext):
            dinv1, info1 = hip_backend.potrf_(a1)
        ext.synchronize()
        assert mixed_launches() == 0                         # the masked stream: none
        assert int(info0.abs().max()) == 0 and int(info1.abs().max()) == 0
        assert torch.equal(torch.tril(a0), torch.tril(a1)) and torch.equal(dinv0, dinv1)
    finally:
        torch.cuda.synchronize()
        hip.hipStreamDestroy(stream)


def test_mixed_phase_batched_factorisation_reproduces_the_lockstep_factors_every_time():
    """``gpk_selftest --batched-stress``: cfg4's 512 x 2048² factorised 40 times by the mixed-phase steps, the XOR fingerprint of EVERY
    lower triangle against one run of the lockstep launches -- the fence-free ordering of solve and update tiles (per-matrix counters,
    all tasks of a matrix on one XCD) gives the same bits every time (300 / 300 in ``profiles/r06_batched_stress.log``)."""
    import subprocess

    exe = os.path.join(ROOT, "stheno_amd", "csrc", "gpk_selftest")
    if not os.path.exists(exe):
        pytest.skip("native self-test not built")
    r = subprocess.run([exe, "--batched-stress", "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and ": 0 differ" in r.stdout, r.stdout[-500:] + r.stderr[-500:]


@pytest.mark.parametrize("b,n", [(64, 512), (70, 640), (96, 1024)])
def test_mixed_phase_batches_give_the_lockstep_factors_bit_for_bit(hip_backend, b, n):
    """The public path: a batch of at least 64 fp32 GPs of an order that is a multiple of 128 is factorised by the mixed-phase steps,
    the same GPs in two batches of fewer than 64 by the lockstep launches -- the same arithmetic in the same order per entry, so the
    FACTORS agree bit for bit; three log-densities against the oracle at 1e-3 (batched semantics: ``stheno/random.py:261,274``,
    ``tests/model/test_cases.py:134-155``)."""
    rng = np.random.default_rng(b + n)
    x = rng.standard_normal((b, n, 3)).astype(np.float32)
    y = rng.standard_normal((b, n, 1)).astype(np.float32)
    tx, ty = torch.as_tensor(x, device="cuda"), torch.as_tensor(y, device="cuda")
    eps0 = st.B.epsilon
    try:
        st.B.epsilon = 1e-6
        f = st.GP(st.EQ())
        fdd = f(tx, 0.1)
        whole = fdd.logpdf(ty)
        h = b // 2
        parts = [f(tx[:h].contiguous(), 0.1), f(tx[h:].contiguous(), 0.1)]
        halves = torch.cat([parts[0].logpdf(ty[:h].contiguous()), parts[1].logpdf(ty[h:].contiguous())])
        # the FACTORS bit for bit (the single-column solves behind the log-density are different kernels for >= 64 and < 64 matrices:
        # their sums run in another order, the log-densities agree to rounding)
        l_whole = fdd.var.chol().lower()
        l_halves = torch.cat([parts[0].var.chol().lower(), parts[1].var.chol().lower()])
        assert torch.equal(l_whole, l_halves)
        assert whole.shape == (b,) and float(((whole - halves).abs() / whole.abs()).max()) <= 2e-6
        for i in (0, b // 3, b - 1):
            ref = O.gp_logpdf([("eq", 1.0, 1.0)], x[i].astype(np.float64), 0.1, y[i].astype(np.float64), eps=1e-6)
            assert abs(float(whole[i]) - ref) <= 1e-3 * abs(ref)
    finally:
        st.B.epsilon = eps0
