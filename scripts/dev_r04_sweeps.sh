#!/bin/bash
# Development aid (round 4): knob sweeps on the dev build.
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
S=./gpk_selftest
{
for v in 512 1099511627776 512 1099511627776; do
  echo "== 64-tile kernels below $v 128-tiles (knob 1): batched factorisation"
  for nbo in 512 1024; do $S --set 1 $v --batched $nbo | tail -1; done
  $S --set 1 $v --gemm f32 16384 2048 512 0 | tail -1
done
} 2>&1 | tee ../../$out/sweep_batched_64_tiles.log
