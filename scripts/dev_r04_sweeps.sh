#!/bin/bash
# Development aid (round 4): knob sweeps on the dev build.
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
S=./gpk_selftest
{
timeout 600 $S > ../../$out/selftest_left.log 2>&1; tail -1 ../../$out/selftest_left.log
for v in 1 0 1 0; do
  echo "== batched factorisation left-looking=$v (knob 44)"
  for nbo in 256 512 1024; do $S --set 44 $v --batched $nbo | tail -1; done
done
} 2>&1 | tee ../../$out/sweep_batched_left_looking.log
cd ../..
python -m pytest tests -q -m gpu -k "batched or config4 or fuzz" 2>&1 | tail -3
python bench.py --workload batched_f32 --steps 10 --warmup 2 --no-cpu-baseline | tail -c 800
