#!/bin/bash
# Development aid (round 4): knob sweeps on the dev build.
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
S=./gpk_selftest
{
for v in 0 1 0 1; do
  echo "== batched 128-tile launches pinned per XCD = $v (knob 45)"
  for nbo in 512 1024; do $S --set 45 $v --batched $nbo | tail -1; done
done
} 2>&1 | tee ../../$out/sweep_batched_xcd_pinning.log
