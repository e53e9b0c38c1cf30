"""Development aid: the two %timeit cases of the reference README (README.md:686-712, hardware unstated there:
219 ms per dense logpdf at N=2000, 9.8 ms per VFE ELBO at N=2000 / M=100) on this path, fp64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stheno_amd as st

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
x_obs = torch.linspace(0, 10, 2000, dtype=torch.float64).to(dev)
x_ind = torch.linspace(0, 10, 100, dtype=torch.float64).to(dev)
st.B.epsilon = 1e-9      # 100 inducing points 0.1 apart under a unit length scale: K_z needs more than the 1e-12 default here
y_obs = torch.randn(2000, 1, generator=g, dtype=torch.float64).to(dev)
prior = st.Measure()
f = st.GP(st.EQ(), measure=prior)
u = st.GP(st.EQ(), measure=prior)


def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    float(out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t1 = timeit(lambda: f(x_obs, 1).logpdf(y_obs))
t2 = timeit(lambda: st.PseudoObs(f(x_ind), f(x_obs, 1), y_obs).elbo(prior))
print(f"dense logpdf N=2000: {1e3 * t1:.2f} ms per call;  VFE ELBO N=2000 M=100: {1e3 * t2:.2f} ms per call")
