"""Development: single-matrix factorisations (pipelined panels; look-ahead with its helper stream; rows under the matrix) on several
concurrent streams, bitwise against the same factorisation run alone."""
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402

from stheno_amd import ops  # noqa: E402

be = ops.get_backend()
terms = ops.KTerms([("eq", 1.0, 1.0)])
g = torch.Generator().manual_seed(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def job(kind, seed):
    gg = torch.Generator().manual_seed(seed)
    if kind == "plain":
        x = torch.randn(3072, 4, generator=gg, dtype=torch.float64).cuda()
        def run():
            a = be.kmat(terms, x, lower=True, diag_add=0.1)
            dinv, info = be.potrf_(a)
            return torch.tril(a), dinv, info
    elif kind == "la":
        x = torch.randn(12288, 4, generator=gg, dtype=torch.float32).cuda()
        def run():
            a = be.kmat(terms, x, lower=True, diag_add=0.1)
            dinv, info, dnb = be.potrf_(a, lookahead_nb=1024, lookahead_sb=512)
            return torch.tril(a), dinv, info, dnb
    else:
        x = torch.randn(4096, 4, generator=gg, dtype=torch.float64).cuda()
        xs = torch.randn(256, 4, generator=gg, dtype=torch.float64).cuda()
        def run():
            buf = torch.empty((4096 + 256, 4096), dtype=torch.float64, device="cuda")
            be.kmat(terms, x, lower=True, diag_add=0.1, out=buf[:4096])
            be.kmat(terms, xs, x, out=buf[4096:])
            dinv, info, _ = be.potrf_rows_(buf)
            return torch.tril(buf[:4096]), buf[4096:].clone(), dinv, info
    return run


jobs = [job("plain", 1), job("plain", 2), job("la", 3), job("rows", 4), job("plain", 5), job("rows", 6)]
ref = [j() for j in jobs]
torch.cuda.synchronize()
S = [torch.cuda.Stream() for _ in jobs]
bad = 0
for it in range(iters):
    cur = torch.cuda.current_stream()
    outs = [None] * len(jobs)
    for i, s in enumerate(S):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs[i] = jobs[i]()
    for s in S:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    for i in range(len(jobs)):
        for k, (a, b) in enumerate(zip(outs[i], ref[i])):
            if not torch.equal(a, b):
                ne = (a != b) | (a != a)
                idx = torch.nonzero(ne)
                print(f"iteration {it} job {i} output {k}: {idx.shape[0]} entries differ, first {idx[0].tolist()} last {idx[-1].tolist()}, NaN {int(torch.isnan(a.double()).sum())}", flush=True)
                bad += 1
                break
print("iterations", iters, "jobs", len(jobs), "mismatches", bad)
