"""Experiment: cfg4 (512 x N=2048 fp32 logpdf) as ONE batch on one stream vs 2 / 4 sub-batches on separate streams
(low-efficiency phases of one sub-batch -- diagonal blocks, K=128 panel work -- under the trailing GEMMs of another)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import stheno_amd as st  # noqa: E402

st.B.epsilon = 1e-6
B, n, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 512), 2048, 3      # (argv[1]: the batch -- 64 / 128 / 256 = an 8 / 4 / 2-GPU rank's shard of cfg4)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, n, d, generator=g, dtype=torch.float32).cuda()
y = torch.randn(B, n, 1, generator=g, dtype=torch.float32).cuda()
f = st.GP(st.EQ())


from stheno_amd import matrix  # noqa: E402

# (round 5: without these two the HOST serialises the sub-batches -- every logpdf reads its NaN flag and its info word before the next
# sub-batch's launches are even enqueued -- and the round-2 run of this script measured nothing but that)
matrix.config.check_nan = False


STREAMS = [torch.cuda.Stream() for _ in range(8)]      # (made once: a fresh stream per call gets a fresh allocator pool -- 8.6 GB of hipMalloc)


def run(parts):
    if parts == 1:
        with st.deferred_checks():
            return f(x, 0.1).logpdf(y)
    streams = STREAMS[:parts]
    step = B // parts
    outs = [None] * parts
    cur = torch.cuda.current_stream()
    with st.deferred_checks():
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs[i] = f(x[i * step:(i + 1) * step], 0.1).logpdf(y[i * step:(i + 1) * step])
        for s in streams:
            cur.wait_stream(s)
    return torch.cat(outs)


ref = None
for parts in ((1, 2, 4, 8, 1, 2, 4, 8, 1, 2, 4) if B >= 512 else (1, 2, 4, 1, 2, 4, 1, 2)):
    try:
        for _ in range(4):
            out = run(parts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = run(parts)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        if ref is None:
            ref = out
        print(f"parts={parts}: {ms:.3f} ms per {B} GPs, max |diff| vs one batch {float((out - ref).abs().max()):.3e}", flush=True)
    except Exception as e:
        torch.cuda.synchronize()
        print(f"parts={parts}: FAILED {repr(e)[:150]}", flush=True)
