#!/bin/bash
# Round 6, pass 3: lag sweep of the mixed-phase batched steps; the fp32 diagonal-block kernel at two workgroups per CU (128 registers, spills) A/B.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass3
mkdir -p $O
cd $R/stheno_amd/csrc
for lag in 192 256 384 512 768 1024; do
  timeout 120 ./gpk_selftest --set 53 1 --set 55 $lag --batched 0 2>&1 | grep "BATCHED potrf" | sed "s/^/[lag=$lag] /" | tee -a $O/batched_lag.log
done
for b in . ab_diag4; do
  for mode in 0 1; do
    timeout 120 $b/gpk_selftest --set 53 $mode --set 55 384 --batched 0 2>&1 | grep "BATCHED potrf\|differing" | sed "s/^/[$b 53=$mode] /" | tee -a $O/batched_diag4.log
  done
done
echo "finished at $SECONDS s"
