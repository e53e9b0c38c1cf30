"""Round 5: batched factorisations on EIGHT concurrent streams must give the single-stream bits.

Found while re-running the two-stream experiment for cfg4: about one 128 x 128 diagonal block in 1e5 came out wrong from local row 80,
column 24 on (a not-positive-definite report, or a log-density off by 6e-5) when 8 sub-batches of 64 matrices were factorised on 8
streams at once -- never on one stream.  ``potrf_diag3_kernel``: the waves that factorise a 16-column micro-panel each repeat the
diagonal tile's arithmetic on their lanes 0..15, i.e. read the diagonal tile's rows from LDS, and wave 0 overwrites those rows with the
factor when it is done; the compiler had sunk the loads of columns 8..15 several pivots into the loop, so a wave that fell behind wave 0
(its SIMD shared with another kernel's waves) read factorised values.  Fixed by reading all rows in front of a workgroup barrier
(``panel_load``, ``gpk_potrf.hip``).  ``scripts/dev_stream_race.py`` is the stage-by-stage hunt; this is the regression test
(reference semantics: the batched computation of ``tests/model/test_cases.py:134-155``, results independent of what runs beside)."""
import pytest
import torch

from stheno_amd import ops

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hip_backend")]


@pytest.mark.parametrize("dtype,n,iters", [(torch.float32, 2048, 250), (torch.float64, 1024, 250)])
def test_batched_factorisations_on_eight_concurrent_streams_give_the_single_stream_bits(dtype, n, iters):
    be = ops.get_backend()
    terms = ops.KTerms([("eq", 1.0, 1.0)])
    g = torch.Generator().manual_seed(0)
    parts, per = 8, 32
    x = torch.randn(parts * per, n, 3, generator=g, dtype=torch.float64).to(dtype).cuda()

    def factor(xs):
        a = be.kmat(terms, xs, lower=True, diag_add=0.1 + 1e-6)
        dinv, info = be.potrf_(a)
        return torch.tril(a), dinv, info

    ref = [factor(x[i * per:(i + 1) * per]) for i in range(parts)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for it in range(iters):
        cur = torch.cuda.current_stream()
        outs = [None] * parts
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs[i] = factor(x[i * per:(i + 1) * per])
        for s in streams:
            cur.wait_stream(s)
        torch.cuda.synchronize()
        for i in range(parts):
            for got, want, what in zip(outs[i], ref[i], ("factor", "block inverses", "info")):
                assert torch.equal(got, want), f"iteration {it}, sub-batch {i}: {what} differs from the single-stream run"
