#!/bin/bash
# Development aid: A/B of the 128-tile k loop (GPK_GEMM_PIPE, gpk_gemm_tile.hpp) on one box.
# `make -C stheno_amd/csrc all && make -C stheno_amd/csrc ab PIPE=0` first; run through gpurun from the repo root.
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
for b in . ab0 . ab0; do
  echo "== build $b"
  $b/gpk_selftest --gemm f64 8192 8192 8192 | tail -2
  $b/gpk_selftest --gemm f32 8192 8192 8192 | tail -2
  $b/gpk_selftest --gemm f64 15360 15360 1024 1 | tail -2
  $b/gpk_selftest --gemm f32 30720 30720 1024 1 | tail -2
  $b/gpk_selftest --gemm f64 8192 2048 8192 64 | tail -2
  $b/gpk_selftest --gemm f32 16384 2048 16384 64 | tail -2
  $b/gpk_selftest --gemm f32 4096 200064 4096 68 | tail -2
done 2>&1 | tee ../../$out/ab_gemm_pipe.log
for b in . ab0; do echo "== build $b"; $b/gpk_selftest --perf-la 16384 2>&1 | grep -E "potrf_f64 n=16384|trail_f64 n=15360|potrf_f32 n=16384"; done 2>&1 | tee ../../$out/ab_perf_la.log
./gpk_selftest > ../../$out/selftest.log 2>&1; tail -3 ../../$out/selftest.log
