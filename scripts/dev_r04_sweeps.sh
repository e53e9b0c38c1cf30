#!/bin/bash
# Development aid (round 4): knob sweeps on the dev build after the GEMM changes.
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
S=./gpk_selftest
{
for v in 1024 512 384 256; do
  echo "== 64-tile kernels below $v 128-tiles (knob 1)"
  $S --set 1 $v --perf-trsm | grep -E "sb=1024 out|sb=256 out|sb=512 out" | awk 'NR%2==0'
  $S --set 1 $v --la-one f64 16384 1024 1 6144 3 | tail -1
  $S --set 1 $v --la-one f32 32768 1024 1 6144 2 | tail -1
  $S --set 1 $v --batched 512 | tail -1
  $S --set 1 $v --profile f64 4096 0 | tail -1
  $S --set 1 $v --profile f64 8192 0 | tail -1
done
} 2>&1 | tee ../../$out/sweep_small_tile.log
