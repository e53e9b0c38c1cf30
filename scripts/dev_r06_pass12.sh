#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass12
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 300 ./gpk_selftest --set 61 1 --potrf 2>&1 | tail -1
for mode in "53 0" "53 1 --set 61 0" "53 1 --set 61 1" "53 1 --set 61 0" "53 1 --set 61 1"; do
  timeout 120 ./gpk_selftest --set $mode --batched 0 2>&1 | grep "BATCHED potrf\|differing" | sed "s/^/[$mode] /" | tee -a $O/batched_diag_inside.log
done
echo "finished at $SECONDS s"
