#!/bin/bash
# Development aid: A/B of two builds of the library on one box (the default dev build against `make ab ABDIR=... ABFLAGS=...`).
# usage (through gpurun, from the repo root): bash scripts/dev_ab_gemm.sh ab_clds0
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
timeout 600 ./gpk_selftest > ../../$out/selftest_ab.log 2>&1; tail -1 ../../$out/selftest_ab.log
for b in . $1 . $1; do
  echo "== build $b"
  $b/gpk_selftest --gemm f64 8192 8192 8192 | tail -1
  $b/gpk_selftest --gemm f64 15360 15360 1024 1 | tail -1
  $b/gpk_selftest --gemm f32 30720 30720 1024 1 | tail -1
  $b/gpk_selftest --gemm f64 8192 2048 8192 64 | tail -1
  $b/gpk_selftest --gemm f32 16384 2048 512 0 | tail -1
  $b/gpk_selftest --gemm f32 16384 2048 128 0 | tail -1
  $b/gpk_selftest --batched 512 | tail -1
  $b/gpk_selftest --la-one f64 16384 1024 1 6144 3 | tail -1
  $b/gpk_selftest --la-one f32 32768 1024 1 6144 2 | tail -1
  $b/gpk_selftest --perf-trsm | grep "sb=1024 out" | tail -2
done 2>&1 | tee ../../$out/ab_$1.log
