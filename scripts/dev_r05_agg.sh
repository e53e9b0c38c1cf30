#!/bin/bash
# Round 5, first GPU pass: correctness of the aggregated trailing updates (native self-test, look-ahead section + the GEMM K = 0 cases),
# then the A/B of the aggregation depth m in ONE process (alternating), then the bench line.  stderr kept.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 400 ./gpk_selftest > $O/selftest_dev.log 2>&1; echo "selftest(dev) rc=$?"; tail -1 $O/selftest_dev.log; grep FAIL $O/selftest_dev.log | head -20
timeout 300 ./gpk_selftest_rel > $O/selftest_rel.log 2>&1; echo "selftest(release) rc=$?"; tail -1 $O/selftest_rel.log; grep FAIL $O/selftest_rel.log | head -20
timeout 120 ./gpk_selftest --perf-agg f64 16384 1024 0 4 0 1 2 3 4 > $O/perf_agg_f64.log 2>&1; grep SUMMARY $O/perf_agg_f64.log
timeout 120 ./gpk_selftest --perf-agg f64 16384 1024 0 3 4096 1 2 4 > $O/perf_agg_f64_tail4096.log 2>&1; grep SUMMARY $O/perf_agg_f64_tail4096.log
timeout 200 ./gpk_selftest --perf-agg f32 32768 1024 512 3 0 1 2 3 4 > $O/perf_agg_f32.log 2>&1; grep SUMMARY $O/perf_agg_f32.log
cd /tmp
timeout 200 python $R/bench.py --steps 20 --warmup 5 2> $O/bench_dense.err | grep "^{" | tail -1 > $O/bench_dense_f64.json; cat $O/bench_dense_f64.json
timeout 200 python $R/bench.py --workload sum_f32 --no-batched-record --no-cpu-baseline 2> $O/bench_sum.err | grep "^{" | tail -1 > $O/bench_sum_f32.json; cat $O/bench_sum_f32.json
echo "finished at $SECONDS s"
