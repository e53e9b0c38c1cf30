"""Round 4: the pipelined k loop of the kernels WITHOUT bounds checks fetches its operands with buffer loads -- a wave-uniform origin
in a descriptor plus 32-bit offsets (``gpk_gemm_tile.hpp``, BUF).  The offsets cover 128 rows of a leading dimension below
``GPK_PIPE_LD_MAX`` = 2^21 elements; the launchers send anything wider to the bounds-checked kernels (per-thread 64-bit pointers).
Checked against torch in fp64 on both sides of the threshold, in every operand layout."""
import pytest
import torch

from stheno_amd import ops

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hip_backend")]
DEV = torch.device("cuda")
LD_MAX = 1 << 21


def _strided(rows, cols, ld, dtype, gen):
    """A (rows, cols) view with leading dimension ``ld`` (the storage is one (rows, ld) block; only the view is initialised)."""
    store = torch.empty(rows, ld, dtype=dtype, device=DEV)
    view = store[:, :cols]
    view.copy_(torch.randn(rows, cols, generator=gen, dtype=torch.float64).to(dtype))
    return view


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("ld", [LD_MAX - 128, LD_MAX, LD_MAX + 256])
@pytest.mark.parametrize("a_kmajor,b_kmajor", [(True, True), (True, False), (False, True), (False, False)])
def test_full_tiles_with_wide_leading_dimensions(dtype, ld, a_kmajor, b_kmajor):
    be = ops.get_backend()
    g = torch.Generator().manual_seed(ld % 1000 + 2 * a_kmajor + b_kmajor)
    M, N, K = 256, 128, 64                       # full 128-tiles, whole k chunks: the kernels without bounds checks below the threshold
    a = _strided(M, K, ld, dtype, g) if a_kmajor else _strided(K, M, ld, dtype, g)
    b = _strided(N, K, ld, dtype, g) if b_kmajor else _strided(K, N, ld, dtype, g)
    out = be.gemm(a, b, a_kmajor=a_kmajor, b_kmajor=b_kmajor)
    a64 = a.double() if a_kmajor else a.double().T
    b64 = b.double() if b_kmajor else b.double().T
    ref = a64 @ b64.T
    tol = 1e-13 if dtype == torch.float64 else 1e-5
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < tol


def test_update_in_place_with_a_wide_leading_dimension():
    """C -= A A^T on the lower triangle (the shape of the Cholesky's trailing update) with ld at the threshold."""
    be = ops.get_backend()
    g = torch.Generator().manual_seed(7)
    n, k, ld = 384, 128, LD_MAX
    a = _strided(n, k, ld, torch.float64, g)
    c = torch.randn(n, n, generator=g, dtype=torch.float64).to(DEV)
    want = c - a @ a.T
    be.gemm(a, a, alpha=-1.0, beta=1.0, out=c, lower_only=True)
    assert float((torch.tril(c) - torch.tril(want)).abs().max()) < 1e-11
