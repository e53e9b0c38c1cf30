#!/bin/bash
# Development aid: A/B of the tile streams (gpk_tune 43; gpk_gemm_stream.hpp) against one-tile-at-a-time launches, same binary, same box.
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
S=./gpk_selftest
{
timeout 600 $S > ../../$out/selftest_stream.log 2>&1; tail -2 ../../$out/selftest_stream.log
for v in 1 0 1 0; do
  echo "== stream=$v"
  $S --set 43 $v --gemm f64 8192 8192 8192 | tail -1
  $S --set 43 $v --gemm f64 15360 15360 1024 1 | tail -1
  $S --set 43 $v --gemm f32 30720 30720 1024 1 | tail -1
  $S --set 43 $v --gemm f64 8192 2048 8192 64 | tail -1
  $S --set 43 $v --gemm f32 16384 2048 512 0 | tail -1
  $S --set 43 $v --gemm f32 16384 2048 128 0 | tail -1
  $S --set 43 $v --batched 512 | tail -1
  $S --set 43 $v --la-one f64 16384 1024 1 6144 3 | tail -1
  $S --set 43 $v --la-one f32 32768 1024 1 6144 2 | tail -1
done
} 2>&1 | tee ../../$out/ab_gemm_stream.log
