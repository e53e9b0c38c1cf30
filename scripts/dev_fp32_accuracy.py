"""fp32 accuracy of the dense path (cfg3 kernel EQ + Linear, D = 4) against fp64 on the same inputs,
for different sizes of the explicitly inverted diagonal blocks of the solves and for the plain / look-ahead
factorisations -- next to what LAPACK in fp32 (NumPy/SciPy float32, the reference's own CPU path at that
precision) gets on the same inputs.  Run on the GPU box:  python scripts/dev_fp32_accuracy.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import scipy.linalg as sla
import torch

import stheno_amd as st
from stheno_amd import B, matrix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ns, d, noise = 2048, 4, 0.1
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, d, device="cuda", generator=g)
xs = torch.randn(ns, d, device="cuda", generator=g)
y = torch.randn(n, 1, device="cuda", generator=g)
k = st.EQ() + st.Linear()
f = st.GP(k)


def run(dtype, eps):
    B.epsilon = eps
    xx, yy, xxs = x.to(dtype), y.to(dtype), xs.to(dtype)
    fd = f(xx, noise)
    lp = fd.logpdf(yy)
    m, v = (f | (fd, yy))(xxs).marginals()
    return float(lp), m.double().cpu().numpy(), v.double().cpu().numpy()


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


lp64, m64, v64 = run(torch.float64, 1e-12)
print(f"N={n}: fp64 logpdf {lp64:.6f}")

# LAPACK in fp32 on the host (the reference's NumPy path in float32)
xn, yn, xsn = x.cpu().numpy().astype(np.float32), y.cpu().numpy().astype(np.float32), xs.cpu().numpy().astype(np.float32)


def kern(a, b):
    d2 = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2 * a @ b.T
    return (np.exp(-0.5 * np.maximum(d2, 0)) + a @ b.T).astype(np.float32)


K = kern(xn, xn) + np.float32(noise + 1e-6) * np.eye(n, dtype=np.float32)
L = sla.cholesky(K, lower=True)
a = sla.solve_triangular(L, yn, lower=True)
lp_np = -0.5 * (2 * np.log(np.diag(L).astype(np.float64)).sum() + n * np.log(2 * np.pi) + float((a.astype(np.float64) ** 2).sum()))
V = sla.solve_triangular(L, kern(xn, xsn), lower=True)
m_np = (V.T @ a)[:, 0].astype(np.float64)
v_np = ((np.exp(0) + (xsn * xsn).sum(1)).astype(np.float32) - (V * V).sum(0)).astype(np.float64)
print(f"LAPACK fp32 (host)      : logpdf {abs(lp_np - lp64) / abs(lp64):.2e}  mean {rel(m_np, m64):.2e}  var {rel(np.maximum(v_np, 0), v64):.2e}")

orig = matrix._solve_block
for la_from, nb in [(0, 0), (1024, 512), (1024, 1024)]:
    matrix.config.potrf_lookahead_from = la_from
    if nb:
        matrix.config.potrf_lookahead_nb = {torch.float64: 1024, torch.float32: nb}
    for sb in (128, 256, 512, 1024):
        matrix._solve_block = (lambda s: (lambda nn, nrhs, fp64=True: s))(sb)
        lp, m, v = run(torch.float32, 1e-6)
        print(f"potrf {'plain' if not la_from else 'look-ahead nb=%d' % nb:20s} solve blocks {sb:4d}: logpdf {abs(lp - lp64) / abs(lp64):.2e}  mean {rel(m, m64):.2e}  var {rel(v, v64):.2e}")
matrix._solve_block = orig
