#!/bin/bash
# Development aid (round 4): knob sweeps on the dev build after the GEMM changes.
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
S=./gpk_selftest
{
for v in 0 1 2 0 1 2; do
  echo "== look-ahead panel GEMM mode $v (knob 10)"
  $S --set 10 $v --la-one f64 16384 1024 1 6144 3 | tail -1
  $S --set 10 $v --la-one f32 32768 1024 1 6144 2 | tail -1
done
} 2>&1 | tee ../../$out/sweep_la_panel_mode.log
cd ../..
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched-record | tail -c 1500
