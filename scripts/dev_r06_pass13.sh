#!/bin/bash
# Round 6, pass 13: the next step's diagonal blocks as four-wave tasks of the mixed-phase launch (knob 61)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass13
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 300 ./gpk_selftest --potrf 2>&1 | tail -1
timeout 300 ./gpk_selftest --potrf 2>&1 | grep FAIL | head
for mode in "53 0" "53 1 --set 61 0" "53 1 --set 61 1" "53 1 --set 61 0" "53 1 --set 61 1"; do
  timeout 120 ./gpk_selftest --set $mode --batched 0 2>&1 | grep "BATCHED potrf\|differing" | sed "s/^/[$mode] /" | tee -a $O/batched_diag_tasks.log
done
timeout 200 ./gpk_selftest --batched-stress 100 | tee -a $O/batched_diag_tasks.log
echo "finished at $SECONDS s"
