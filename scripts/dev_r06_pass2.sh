#!/bin/bash
# Round 6, pass 2: the mixed-phase batched steps -- correctness (self-test factorisation cases, bitwise A/B against the lockstep launches) and time.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass2
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 300 ./gpk_selftest --potrf > $O/selftest_potrf.log 2>&1; echo "selftest --potrf rc=$? $(tail -1 $O/selftest_potrf.log)"; grep FAIL $O/selftest_potrf.log | head -20
for mode in 0 1; do
  timeout 120 ./gpk_selftest --set 53 $mode --batched 0 2>&1 | tee -a $O/batched_ab.log | sed "s/^/[53=$mode] /"
done
for lag in 32 96 256; do
  timeout 120 ./gpk_selftest --set 53 1 --set 55 $lag --batched 0 2>&1 | grep "BATCHED potrf" | sed "s/^/[lag=$lag] /" | tee -a $O/batched_ab.log
done
echo "finished at $SECONDS s"
