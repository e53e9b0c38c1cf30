"""List the host<->device synchronisation points of one dense eval (logpdf + condition + marginals): torch's sync debug mode warns at
every implicit synchronisation with a Python stack.  usage: python scripts/dev_find_host_syncs.py [N] [deferred]"""
import sys
import warnings

import torch

import stheno_amd as st

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
deferred = len(sys.argv) > 2
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.rand(n, 8, generator=g, dtype=torch.float64).to(dev)
y = torch.randn(n, 1, generator=g, dtype=torch.float64).to(dev)
xs = torch.rand(512, 8, generator=g, dtype=torch.float64).to(dev)


def step():
    f = st.GP(st.EQ())
    fdd = f(x, 0.1)
    lp = fdd.logpdf(y)
    post = f | (fdd, y)
    mean, var = post(xs).marginals()
    return lp, mean, var


step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as rec:
    warnings.simplefilter("always")
    if deferred:
        with st.deferred_checks():
            step()
    else:
        step()
torch.cuda.set_sync_debug_mode("default")
print(f"{len(rec)} synchronising calls in one eval")
import traceback
for w in rec:
    print("-", str(w.message).split("\n")[0][:100], "@", f"{w.filename.split('/')[-1]}:{w.lineno}")
