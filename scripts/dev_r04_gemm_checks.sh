#!/bin/bash
# Development aid (round 4): the GEMM shapes of the four configurations + the per-tile time stamps of the persistent update.
out=gpurun_out/r04; mkdir -p $out
cd stheno_amd/csrc
S=./gpk_selftest
{
timeout 600 $S > ../../$out/selftest.log 2>&1; tail -1 ../../$out/selftest.log
$S --gemm f64 8192 8192 8192 | tail -1
$S --gemm f64 15360 15360 1024 1 | tail -1
$S --gemm f32 30720 30720 1024 1 | tail -1
$S --gemm f64 8192 2048 8192 64 | tail -1
$S --gemm f32 16384 2048 512 0 | tail -1
$S --gemm f32 16384 2048 128 0 | tail -1
$S --gemm f32 4096 200000 4096 68 | tail -1
$S --gemm f64 14464 14464 1024 1 | tail -1
$S --batched 512 | tail -1
$S --la-one f64 16384 1024 1 6144 3 | tail -1
$S --la-one f32 32768 1024 1 6144 2 | tail -1
$S --tileprof 1024 1 2
$S --tileprof 256 1 2
} 2>&1 | tee ../../$out/gemm_checks.log
