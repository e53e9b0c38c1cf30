"""Development aid: per-step wall time + allocator activity of a bench workload."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stheno_amd as st
from bench import make_inputs, make_step

name = sys.argv[1] if len(sys.argv) > 1 else "batched_f32"
dev = torch.device("cuda")
w, t = make_inputs(name, dev)
if w["dtype"] == "f32":
    st.B.epsilon = 1e-6
step = make_step(name, w, t)
keep = None
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    keep = step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ms = torch.cuda.memory_stats()
    print(f"step {i}: {1e3*(t1-t0):8.2f} ms  device_allocs={ms['num_device_alloc']} device_frees={ms['num_device_free']} reserved={ms['reserved_bytes.all.current']/2**30:.2f} GiB allocated={ms['allocated_bytes.all.current']/2**30:.2f} GiB")
