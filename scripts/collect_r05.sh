#!/bin/bash
# Round-5 collection on the GPU box (from the repo root through gpurun): self-tests, the whole GPU suite, bench lines + rocprofv3
# kernel stats + PMC traffic + SQ counters of the four workloads, kernel sequences of one dense eval in both call orders, the
# look-ahead overlap trace, native micro-benchmarks, the one-rank RCCL record, the full-size CPU baselines.  Every step keeps its
# stderr (r04's scripts sent it to /dev/null and an empty record went unnoticed).  Everything lands in gpurun_out/r05_profiles/.
# usage: BUDGET=1500 bash scripts/collect_r05.sh        (seconds of run time this call may use; steps are skipped when it runs out)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05_profiles
mkdir -p $O
BUDGET=${BUDGET:-1500}
left() { echo $((BUDGET - SECONDS)); }
cd $R/stheno_amd/csrc
timeout 500 ./gpk_selftest > $O/r05_selftest.log 2>&1; echo "selftest(dev) rc=$? $(tail -1 $O/r05_selftest.log)"
timeout 400 ./gpk_selftest_rel > $O/r05_selftest_release.log 2>&1; echo "selftest(release) rc=$? $(tail -1 $O/r05_selftest_release.log)"
grep -q "fail=0" $O/r05_selftest.log && grep -q "fail=0" $O/r05_selftest_release.log || { echo "SELFTEST FAILED"; grep FAIL $O/r05_selftest*.log | head -20; exit 1; }
cd $R
timeout 900 python -m pytest tests -m gpu -q -s 2> $O/r05_pytest_gpu.stderr.log | grep -E "ACHIEVED|passed|failed|FAILED|Error" | tee $O/r05_pytest_gpu.log | tail -4
[ -f gpurun_out/r05/achieved_errors.json ] && cp gpurun_out/r05/achieved_errors.json $O/r05_achieved_errors.json
for f in r05_bench_batched_f32_rccl_1rank.json r05_bench_batched_f32_rccl_1rank.stderr.log; do [ -f gpurun_out/r05/$f ] && cp gpurun_out/r05/$f $O/$f; done
grep -q " passed" $O/r05_pytest_gpu.log && ! grep -q "failed" $O/r05_pytest_gpu.log || { echo "PYTEST FAILED"; }
echo "tests done at $SECONDS s"
cd /tmp
for w in dense_f64 sum_f32 batched_f32 sparse_f32; do
  [ $(left) -lt 200 ] && { echo "skipping $w: $(left) s left"; continue; }
  if [ "$w" = dense_f64 ]; then
    timeout 300 python $R/bench.py --steps 20 --warmup 5 2> $O/r05_bench_$w.stderr.log | grep "^{" | tail -1 > $O/r05_bench_$w.json
    timeout 200 python $R/bench.py --steps 20 --warmup 5 --order logpdf-first --no-cpu-baseline --no-batched-record 2>> $O/r05_bench_$w.stderr.log | grep "^{" | tail -1 > $O/r05_bench_${w}_logpdf_first.json
  else
    timeout 300 python $R/bench.py --workload $w --no-batched-record 2> $O/r05_bench_$w.stderr.log | grep "^{" | tail -1 > $O/r05_bench_$w.json
  fi
  [ -s $O/r05_bench_$w.json ] || echo "NO BENCH LINE for $w -- see $O/r05_bench_$w.stderr.log"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-batched-record > $O/r05_stats_$w.log 2>&1
  F=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/r05_bench_${w}_kernel_stats.csv || echo "NO KERNEL STATS for $w -- see $O/r05_stats_$w.log"
  if [ "$w" = dense_f64 ]; then
    T=$(find $O/stats_$w -name "*kernel_trace.csv" | head -1)
    [ -n "$T" ] && python $R/scripts/dev_trace_sequence.py $T kmat 2 > $O/r05_dense_f64_kernel_sequence.txt 2>&1
    [ -n "$T" ] && python $R/scripts/dev_trace_overlap.py $T > $O/r05_dense_f64_lookahead_overlap.txt 2>&1
  fi
  rm -rf $O/stats_$w
  echo "$w bench + stats done at $SECONDS s"
done
for w in dense_f64 batched_f32 sum_f32 sparse_f32; do
  [ $(left) -lt 260 ] && { echo "skipping pmc/sq $w: $(left) s left"; continue; }
  timeout 200 python $R/scripts/collect_pmc.py $w $O/r05_pmc_$w.json > $O/r05_pmc_$w.log 2>&1 || echo "pmc $w failed -- see $O/r05_pmc_$w.log"
  timeout 200 python $R/scripts/collect_sq.py $w $O/r05_sq_$w.json > $O/r05_sq_$w.summary.log 2>&1 || echo "sq $w failed -- see $O/r05_sq_$w.json.log"
  echo "$w pmc + sq done at $SECONDS s"
done
cd $R/stheno_amd/csrc
[ $(left) -gt 100 ] && timeout 90 ./gpk_selftest --perf-rows f64 16384 2048 1024 0 3 > $O/r05_native_perf_rows.log 2>&1
[ $(left) -gt 100 ] && timeout 90 ./gpk_selftest --perf-rows f32 32768 2048 1024 512 2 >> $O/r05_native_perf_rows.log 2>&1
[ $(left) -gt 100 ] && timeout 90 ./gpk_selftest --perf-trsm > $O/r05_native_perf_trsm.log 2>&1
[ $(left) -gt 100 ] && timeout 90 ./gpk_selftest --perf-la > $O/r05_native_perf_lookahead.log 2>&1
[ $(left) -gt 100 ] && timeout 90 ./gpk_selftest --perf-kmat > $O/r05_native_perf_kmat.log 2>&1
cd /tmp
for w in sum_f32 sparse_f32 dense_f64; do
  [ $(left) -lt 320 ] && { echo "skipping full-size cpu baseline $w: $(left) s left"; continue; }
  timeout 300 python $R/bench.py --workload $w --cpu-baseline-full 2> $O/r05_cpu_baseline_full_$w.stderr.log | grep "^{" | tail -1 > $O/r05_cpu_baseline_full_$w.json
  [ -s $O/r05_cpu_baseline_full_$w.json ] || echo "NO full-size cpu baseline for $w -- see $O/r05_cpu_baseline_full_$w.stderr.log"
  echo "$w cpu baseline done at $SECONDS s"
done
ls -la $O | head -80
echo "finished at $SECONDS s"
