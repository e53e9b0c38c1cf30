"""The factorisation's inter-workgroup waits under CONTENTION: the pipelined panel kernel
(``potrf_pipe_kernel``: a chain workgroup and task-queue workers that poll each other's flag words) and the look-ahead (chain on the
CU-masked helper stream beside the persistent update) are run while a SECOND PROCESS keeps every CU of the same GPU busy with large
matrix products.  Nothing may give up waiting (``info = -1``), and the factors must be the ones LAPACK computes.
Reference semantics: ``B.cholesky`` under ``stheno/random.py:274-276``."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

import stheno_amd as st
from stheno_amd import matrix, ops

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hip_backend")]
DEV = torch.device("cuda")

HOG = r"""
import sys, time, torch
a = torch.randn(6144, 6144, device="cuda")
b = torch.randn(6144, 6144, device="cuda")
torch.cuda.synchronize()
print("hog running", flush=True)
t0 = time.time()
n = 0
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(8):
        c = a @ b
    torch.cuda.synchronize()
    n += 8
print("hog done", n, flush=True)
"""


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_factorisations_beside_a_process_that_saturates_the_cus(dtype):
    hog = subprocess.Popen([sys.executable, "-c", HOG, "25"], stdout=subprocess.PIPE, text=True, env=dict(os.environ))
    try:
        assert "hog running" in hog.stdout.readline()
        be = ops.get_backend()
        t_end = time.time() + 12.0
        rounds = 0
        cases = [(4096, 0), (8192 + 64, 0), (12288, 1024)]          # one pipelined panel / several panels with fill tiles / look-ahead
        mats = {}
        for n, _ in cases:
            g = torch.Generator().manual_seed(n)
            x = torch.randn(n, 6, generator=g, dtype=torch.float64).to(dtype).to(DEV)
            a = st.EQ().pairwise(x, None)
            a.diagonal().add_(0.2)
            mats[n] = a
        ref = np.linalg.cholesky(mats[4096].double().cpu().numpy())
        while time.time() < t_end or rounds < 2:
            for n, la in cases:
                m = mats[n].clone()
                if la:
                    sb = matrix.config.potrf_lookahead_inv[dtype]
                    _, info, _ = be.potrf_(m, 0, lookahead_nb=la, lookahead_sb=min(la, sb))
                else:
                    _, info = be.potrf_(m, 0)
                code = int(info.max())               # (host read: the factorisation has finished)
                assert int(info.min()) != -1 and code == 0, (n, code)      # nobody stopped waiting, positive definite
                if n == 4096:
                    err = np.max(np.abs(torch.tril(m).double().cpu().numpy() - ref)) / np.max(np.abs(ref))
                    assert err < (1e-11 if dtype == torch.float64 else 3e-4), err
            rounds += 1
        assert hog.poll() is None, "the competing process ended before the factorisations did: no contention was exercised"
    finally:
        hog.kill()
        hog.wait()
