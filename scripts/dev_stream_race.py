"""Development: which stage of the batched log-density differs when 8 sub-batches run on 8 concurrent streams?  Each stage is compared
BITWISE with its single-stream result (same sub-batch, same kernels)."""
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402

import stheno_amd as st  # noqa: E402
from stheno_amd import ops  # noqa: E402

be = ops.get_backend()
terms = ops.KTerms([("eq", 1.0, 1.0)])
g = torch.Generator().manual_seed(0)
B, n, d, parts = 512, 2048, 3, 8
x = torch.randn(B, n, d, generator=g, dtype=torch.float32).cuda()
y = torch.randn(B, n, 1, generator=g, dtype=torch.float32).cuda()
step = B // parts


def stages(xs, ys):
    a = be.kmat(terms, xs, lower=True, diag_add=0.1 + 1e-6)
    k0 = a.clone()
    dinv, info = be.potrf_(a)
    v = be.tri_solve_(a, dinv, 128, ys.clone())
    ld = be.logdet_chol(a)
    _, ss = be.colreduce(v, want_ss=True)
    return dict(kmat=k0, L=torch.tril(a), dinv=dinv, v=v, logdet=ld, ss=ss, info=info)


ref = [stages(x[i * step:(i + 1) * step], y[i * step:(i + 1) * step]) for i in range(parts)]
torch.cuda.synchronize()
S = [torch.cuda.Stream() for _ in range(parts)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for it in range(iters):
    cur = torch.cuda.current_stream()
    outs = [None] * parts
    for i, s in enumerate(S):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs[i] = stages(x[i * step:(i + 1) * step], y[i * step:(i + 1) * step])
    for s in S:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    for i in range(parts):
        rep = []
        for key in ("kmat", "L", "dinv", "v", "logdet", "ss", "info"):
            a, b = outs[i][key], ref[i][key]
            if not torch.equal(a, b):
                ne = (a != b) | (a != a)
                idx = torch.nonzero(ne)
                msg = f"{key}: {idx.shape[0]} entries differ, NaN in out {int(torch.isnan(a.float()).sum())} in ref {int(torch.isnan(b.float()).sum())}, first {idx[0].tolist()} last {idx[-1].tolist()}"
                if key == "L":
                    m = idx[0, 0].item()
                    d2 = ne[m]
                    rows = torch.nonzero(d2.any(1)).flatten()
                    cols = torch.nonzero(d2.any(0)).flatten()
                    msg += f" | matrix {m}: rows {rows[0].item()}..{rows[-1].item()} ({rows.numel()}), cols {cols[0].item()}..{cols[-1].item()} ({cols.numel()}); matrices affected {torch.nonzero(ne.flatten(1).any(1)).flatten().tolist()}"
                    blk = d2[:128, :128]
                    msg += f" | inside the first diagonal block: {int(blk.sum())} entries, first {torch.nonzero(blk)[0].tolist() if blk.any() else None}; sample out/ref at first: {float(a[tuple(idx[0].tolist())])} / {float(b[tuple(idx[0].tolist())])}"
                rep.append(msg)
        if rep:
            bad += 1
            print(f"iteration {it} sub-batch {i}: " + " || ".join(rep), flush=True)
print("iterations", iters, "mismatching sub-batches", bad)
