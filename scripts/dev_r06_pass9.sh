#!/bin/bash
# Round 6, pass 9: the persistent update's tiles claimed in chunks of 64 per XCD (knob 58) -- correctness, time, fabric-side traffic.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass9
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 300 ./gpk_selftest --set 58 1 --set 59 0 --lookahead 2>&1 | tail -2
timeout 300 ./gpk_selftest --set 58 1 --set 59 0 --rows 2>&1 | tail -2
for rep in 1 2; do
for c in 0 1; do
  timeout 200 ./gpk_selftest --set 58 $c --perf-rows f64 16384 2048 1024 0 3 2>&1 | grep "round 3" | sed "s/^/[chunks=$c] /" | tee -a $O/ab_chunks_f64.log
done
done
for c in 0 1; do
  timeout 200 ./gpk_selftest --set 58 $c --perf-rows f32 32768 2048 1024 512 2 2>&1 | grep "round 2" | sed "s/^/[chunks=$c] /" | tee -a $O/ab_chunks_f32.log
done
cd /tmp
for c in 0 1; do
  timeout 300 python $R/scripts/pmc_native.py $O/pmc_chunks$c.json -- $R/stheno_amd/csrc/gpk_selftest --set 58 $c --perf-rows f64 16384 2048 1024 0 1 2>&1 | sed "s/^/[chunks=$c] /" | tee -a $O/pmc_chunks.log
done
echo "finished at $SECONDS s"
