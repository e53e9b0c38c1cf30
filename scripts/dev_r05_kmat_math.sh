#!/bin/bash
# Round 5: fp64 kernel-matrix math (table-based exp, one Goldschmidt step in the sqrt, Horner-form Matern polynomials) against
# round 4's (make ab ABDIR=ab_r4math ABFLAGS=-DGPK_KMAT_R4_MATH=1), same box, same process order: accuracy, then time.
cd stheno_amd/csrc
mkdir -p ../../gpurun_out
{
echo "== round 4 math: accuracy"; ./ab_r4math/gpk_selftest --kmat | grep -E "elementwise|SUMMARY|FAIL"
echo "== round 5 math: accuracy"; ./gpk_selftest --kmat | grep -E "elementwise|SUMMARY|FAIL"
for r in 1 2; do
echo "== round 4 math: time (pass $r)"; ./ab_r4math/gpk_selftest --perf-kmat | grep "row-band"
echo "== round 5 math: time (pass $r)"; ./gpk_selftest --perf-kmat | grep "row-band"
done
} > ../../gpurun_out/r05_ab_kmat_math.log 2>&1
tail -60 ../../gpurun_out/r05_ab_kmat_math.log
