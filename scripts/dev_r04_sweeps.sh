#!/bin/bash
# Development aid (round 4): knob sweeps on the dev build after the GEMM changes -- plain-tail length of the look-ahead, TRSM block sizes.
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
S=./gpk_selftest
{
for t in 4096 5120 6144 7168 8192; do echo "== tail rows $t"; $S --set 9 $t --la-one f64 16384 1024 1 6144 4 | tail -2; done
for t in 6144 8192 10240; do echo "== tail rows $t (fp32 N=32768)"; $S --set 9 $t --la-one f32 32768 1024 1 6144 2 | tail -1; done
for m in 1024 2048 4096; do echo "== min overlap rows $m"; $S --set 6 $m --la-one f64 16384 1024 1 6144 3 | tail -1; done
$S --perf-trsm
} 2>&1 | tee ../../$out/sweeps.log
