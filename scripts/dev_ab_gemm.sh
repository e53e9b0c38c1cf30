#!/bin/bash
# Development aid: A/B of two builds of the library on one box (the default dev build against `make ab ABDIR=... ABFLAGS=...`).
# usage (through gpurun, from the repo root): bash scripts/dev_ab_gemm.sh ab_p64off
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
timeout 600 ./gpk_selftest > ../../$out/selftest_ab.log 2>&1; tail -1 ../../$out/selftest_ab.log
for b in . $1 . $1; do
  echo "== build $b"
  $b/gpk_selftest --gemm f64 2048 2048 2048 64 | tail -1
  $b/gpk_selftest --gemm f64 1024 2048 1024 64 | tail -1
  $b/gpk_selftest --gemm f32 4096 2048 4096 64 | tail -1
  $b/gpk_selftest --gemm f64 8192 1024 1024 0 | tail -1
  $b/gpk_selftest --gemm f32 16384 2048 128 0 | tail -1
  $b/gpk_selftest --batched 512 | tail -1
  $b/gpk_selftest --la-one f64 16384 1024 1 6144 3 | tail -1
  $b/gpk_selftest --la-one f32 32768 1024 1 6144 2 | tail -1
  $b/gpk_selftest --profile f64 4096 0 | tail -1
  $b/gpk_selftest --profile f64 8192 0 | tail -1
  $b/gpk_selftest --perf-trsm | grep "sb=1024 out" | awk 'NR%2==0'
done 2>&1 | tee ../../$out/ab_$1.log
