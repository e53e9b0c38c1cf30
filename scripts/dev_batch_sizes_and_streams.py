"""Development check: the batched log-density over sub-batches of several sizes, and over 2 / 4 / 8 sub-batches on concurrent streams,
repeated -- results must equal the one-batch run whatever runs beside what."""
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402

import stheno_amd as st  # noqa: E402
from stheno_amd import matrix  # noqa: E402

st.B.epsilon = 1e-6
g = torch.Generator().manual_seed(0)
B, n, d = 512, 2048, 3
x = torch.randn(B, n, d, generator=g, dtype=torch.float32).cuda()
y = torch.randn(B, n, 1, generator=g, dtype=torch.float32).cuda()
f = st.GP(st.EQ())
ref = f(x, 0.1).logpdf(y)
for bs in (64, 128, 256, 32, 96):
    outs = [f(x[i:i + bs], 0.1).logpdf(y[i:i + bs]) for i in range(0, B - bs + 1, bs)]
    o = torch.cat(outs)
    print("batch", bs, "max rel diff", float(((o - ref[:o.shape[0]]).abs() / ref[:o.shape[0]].abs()).max()))
S = [torch.cuda.Stream() for _ in range(8)]
matrix.config.check_nan = False
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    for parts in (8, 2, 4, 8, 1):
        cur = torch.cuda.current_stream()
        step = B // parts
        outs = [None] * parts
        try:
            with st.deferred_checks():
                if parts == 1:
                    outs[0] = f(x, 0.1).logpdf(y)
                for i, s in enumerate(S[:parts] if parts > 1 else []):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        outs[i] = f(x[i * step:(i + 1) * step], 0.1).logpdf(y[i * step:(i + 1) * step])
                for s in S[:parts]:
                    cur.wait_stream(s)
            o = torch.cat(outs)
            torch.cuda.synchronize()
            print("iteration", it, "parts", parts, "max rel diff", float(((o - ref).abs() / ref.abs()).max()))
        except Exception as e:
            torch.cuda.synchronize()
            print("iteration", it, "parts", parts, "FAILED", repr(e)[:160])
