"""A/B of the width of the explicit inverses in the fp32 look-ahead (outer blocks stay 1024 columns) on the cfg3 bench:
python scripts/dev_ab_fp32_inverse_width.py <repo root> <512|256|1024>.  Measured (ms per eval): 512: 121.4 / 122.4, 256: 123.6 / 123.6."""
import sys, torch
sys.path.insert(0, sys.argv[1])
from stheno_amd import matrix
matrix.config.potrf_lookahead_inv[torch.float32] = int(sys.argv[2])
import bench
sys.argv = ["bench.py", "--workload", "sum_f32", "--no-cpu-baseline", "--steps", "10", "--warmup", "2"]
bench.main()
