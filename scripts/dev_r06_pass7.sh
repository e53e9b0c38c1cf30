#!/bin/bash
# Round 6, pass 7: after the publication-barrier fix (gpk_barrier_stores_done): both self-tests, the single-matrix factorisation times that go through pipe_publish, the batched A/B.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass7
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 500 ./gpk_selftest > $O/selftest.log 2>&1; echo "selftest(dev) rc=$? $(tail -1 $O/selftest.log)"
timeout 400 ./gpk_selftest_rel > $O/selftest_release.log 2>&1; echo "selftest(release) rc=$? $(tail -1 $O/selftest_release.log)"
grep FAIL $O/selftest*.log | head
for i in 1 2 3; do timeout 200 ./gpk_selftest --potrf 2>&1 | tail -1; done
timeout 200 ./gpk_selftest --perf-pipe > $O/perf_pipe.log 2>&1; grep -i "perf" $O/perf_pipe.log | head -30
timeout 200 ./gpk_selftest --perf-rows f64 16384 2048 1024 0 3 2>&1 | tail -4
for mode in "53 0" "53 1"; do
  timeout 120 ./gpk_selftest --set $mode --batched 0 2>&1 | grep "BATCHED potrf\|differing" | sed "s/^/[$mode] /" | tee -a $O/batched_ab.log
done
echo "finished at $SECONDS s"
