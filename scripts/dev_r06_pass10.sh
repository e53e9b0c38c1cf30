#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass10
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 300 ./gpk_selftest --set 60 1 --potrf 2>&1 | tail -1
for ll in 0 1; do for nbo in 0 1024; do
  timeout 120 ./gpk_selftest --set 53 1 --set 60 $ll --batched $nbo 2>&1 | grep "BATCHED potrf\|differing" | sed "s/^/[left=$ll nbo=$nbo] /" | tee -a $O/batched_left.log
done; done
echo "finished at $SECONDS s"
