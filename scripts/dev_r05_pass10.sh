#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_round5_rows.py tests/test_round4_goldens.py tests/test_round4_fused_cols.py -q -k "pseudo or config5 or fused" 2>&1 | tail -6
cd /tmp
for rep in 1 2; do
for flag in True False; do
python - $flag <<PY
import sys, json, subprocess
sys.path.insert(0, "$R")
import torch, time
import stheno_amd as st
from stheno_amd import matrix
from bench import make_inputs, make_step
matrix.config.pseudo_padded_transposed = (sys.argv[1] == "True")
st.B.epsilon = 1e-6
w, t = make_inputs("sparse_f32", torch.device("cuda"))
step = make_step("sparse_f32", w, t)
for _ in range(3): keep = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): keep = step()
torch.cuda.synchronize()
print("sparse_f32 padded_transposed =", sys.argv[1], ":", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms per step, elbo", float(keep))
PY
done
done 2>&1 | grep -v amdgpu.ids | tee $O/ab_sparse_padded_transposed.log
echo "finished at $SECONDS s"
