#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass5
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 120 ./gpk_selftest --set 53 0 --batched 0 2>&1 | grep "BATCHED potrf" | sed "s/^/[lockstep] /" | tee -a $O/batched_mix.log
for opts in 0 4 1; do
 for lag in 32 96 256 512; do
  timeout 120 ./gpk_selftest --set 53 3 --set 55 $lag --set 56 $opts --batched 0 2>&1 | grep "BATCHED potrf\|differing" | tail -2 | sed "s/^/[mix opts=$opts lag=$lag] /" | tee -a $O/batched_mix.log
 done
done
echo "finished at $SECONDS s"
