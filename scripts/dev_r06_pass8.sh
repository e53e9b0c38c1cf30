#!/bin/bash
# Round 6, pass 8: padded rows path on HIP; fp32 pseudo-point margin with K_z evaluated in fp64 and rounded once.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass8
mkdir -p $O
timeout 900 python -m pytest tests/test_round5_rows.py tests/test_round5_host.py tests/test_round6_evidence.py -m gpu -q -x 2>&1 | tail -8
echo "== K_z in fp64, rounded once" | tee $O/sparse_fp32_margin.log
timeout 300 python scripts/dev_r05_sparse_fp32_margin.py 2>&1 | grep sparse_ | tee -a $O/sparse_fp32_margin.log
echo "== K_z evaluated in fp32 (round 5)" | tee -a $O/sparse_fp32_margin.log
timeout 300 python -c "
import runpy, sys
from stheno_amd import matrix
matrix.config.fp64_build_max_order = 0
sys.argv=['x']
runpy.run_path('scripts/dev_r05_sparse_fp32_margin.py', run_name='__main__')" 2>&1 | grep sparse_ | tee -a $O/sparse_fp32_margin.log
echo "finished at $SECONDS s"
