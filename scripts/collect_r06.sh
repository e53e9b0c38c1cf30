#!/bin/bash
# Round-6 collection on the GPU box (from the repo root through gpurun).  Everything lands in gpurun_out/r06_profiles/; every step keeps its stderr.
# usage: BUDGET=2400 STAGE=all bash scripts/collect_r06.sh     (STAGE: tests | bench | counters | native | all)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_profiles
mkdir -p $O
BUDGET=${BUDGET:-2400}
STAGE=${STAGE:-all}
left() { echo $((BUDGET - SECONDS)); }
want() { [ "$STAGE" = all ] || [ "$STAGE" = "$1" ]; }
if want tests; then
  cd $R/stheno_amd/csrc
  timeout 500 ./gpk_selftest > $O/r06_selftest.log 2>&1; echo "selftest(dev) rc=$? $(tail -1 $O/r06_selftest.log)"
  timeout 400 ./gpk_selftest_rel > $O/r06_selftest_release.log 2>&1; echo "selftest(release) rc=$? $(tail -1 $O/r06_selftest_release.log)"
  grep -q "fail=0" $O/r06_selftest.log && grep -q "fail=0" $O/r06_selftest_release.log || { echo "SELFTEST FAILED"; grep FAIL $O/r06_selftest*.log | head -20; }
  cd $R
  timeout 1200 python -m pytest tests -m gpu -q -s 2> $O/r06_pytest_gpu.stderr.log | grep -E "ACHIEVED|passed|failed|FAILED|Error" | tee $O/r06_pytest_gpu.log | tail -4
  [ -f gpurun_out/r05/achieved_errors.json ] && cp gpurun_out/r05/achieved_errors.json $O/r06_achieved_errors.json
  [ -f gpurun_out/r06/achieved.json ] && cp gpurun_out/r06/achieved.json $O/r06_achieved.json
  for f in r05_bench_batched_f32_rccl_1rank.json r05_bench_batched_f32_rccl_1rank.stderr.log; do [ -f gpurun_out/r05/$f ] && cp gpurun_out/r05/$f $O/${f/r05_/r06_}; done
  echo "tests done at $SECONDS s"
fi
cd /tmp
if want bench; then
  for w in dense_f64 sum_f32 batched_f32 sparse_f32; do
    [ $(left) -lt 200 ] && { echo "skipping $w: $(left) s left"; continue; }
    if [ "$w" = dense_f64 ]; then
      timeout 300 python $R/bench.py --steps 20 --warmup 5 2> $O/r06_bench_$w.stderr.log | grep "^{" | tail -1 > $O/r06_bench_$w.json
    else
      timeout 300 python $R/bench.py --workload $w --no-batched-record 2> $O/r06_bench_$w.stderr.log | grep "^{" | tail -1 > $O/r06_bench_$w.json
    fi
    [ -s $O/r06_bench_$w.json ] || echo "NO BENCH LINE for $w -- see $O/r06_bench_$w.stderr.log"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-batched-record > $O/r06_stats_$w.log 2>&1
    F=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/r06_bench_${w}_kernel_stats.csv || echo "NO KERNEL STATS for $w -- see $O/r06_stats_$w.log"
    if [ "$w" = dense_f64 ]; then
      T=$(find $O/stats_$w -name "*kernel_trace.csv" | head -1)
      [ -n "$T" ] && python $R/scripts/dev_trace_sequence.py $T kmat 2 > $O/r06_dense_f64_kernel_sequence.txt 2>&1
    fi
    rm -rf $O/stats_$w
    echo "$w bench + stats done at $SECONDS s"
  done
  # an independent witness of the clock and the power under the two headline workloads, and under the bare MFMA stream
  timeout 200 python $R/scripts/clock_trace.py $O/r06_clock_trace_dense_f64.json -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-batched-record 2> $O/r06_clock_trace.stderr.log | cut -c1-600
  timeout 200 python $R/scripts/clock_trace.py $O/r06_clock_trace_batched_f32.json -- python $R/bench.py --workload batched_f32 --steps 100 --warmup 5 --no-cpu-baseline 2>> $O/r06_clock_trace.stderr.log | cut -c1-600
  timeout 200 python $R/scripts/clock_trace.py $O/r06_clock_trace_mfma_peak.json -- python -c "
import ctypes, torch, json, sys
sys.path.insert(0, '$R')
from stheno_amd import _native
lib = _native.load(); torch.zeros(1, device='cuda')
out = {}
for dt, name in ((1, 'f64'), (0, 'f32')):
    for waves in (1, 2):
        v = [ctypes.c_double() for _ in range(4)]
        lib.gpk_mfma_peak(dt, 1500.0, waves, *[ctypes.byref(x) for x in v], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        out['%s_%dw' % (name, waves)] = dict(tflops=v[0].value, ms=v[1].value, shader_clock_mhz=v[2].value, issue_eff=v[3].value)
print(json.dumps(out))" 2>> $O/r06_clock_trace.stderr.log | cut -c1-900
fi
if want counters; then
  for w in dense_f64 batched_f32 sum_f32 sparse_f32; do
    [ $(left) -lt 260 ] && { echo "skipping pmc/sq $w: $(left) s left"; continue; }
    timeout 200 python $R/scripts/collect_pmc.py $w $O/r06_pmc_$w.json > $O/r06_pmc_$w.log 2>&1 || echo "pmc $w failed -- see $O/r06_pmc_$w.log"
    timeout 200 python $R/scripts/collect_sq.py $w $O/r06_sq_$w.json > $O/r06_sq_$w.summary.log 2>&1 || echo "sq $w failed -- see $O/r06_sq_$w.json.log"
    echo "$w pmc + sq done at $SECONDS s"
  done
fi
if want native; then
  cd $R/stheno_amd/csrc
  [ $(left) -gt 100 ] && timeout 90 ./gpk_selftest --perf-rows f64 16384 2048 1024 0 3 > $O/r06_native_perf_rows.log 2>&1
  [ $(left) -gt 100 ] && timeout 90 ./gpk_selftest --perf-rows f32 32768 2048 1024 512 2 >> $O/r06_native_perf_rows.log 2>&1
  [ $(left) -gt 100 ] && timeout 90 ./gpk_selftest --perf-kmat > $O/r06_native_perf_kmat.log 2>&1
  [ $(left) -gt 100 ] && timeout 90 ./gpk_selftest --perf-pipe > $O/r06_native_perf_pipelined_panel.log 2>&1
  for mode in 0 1; do [ $(left) -gt 60 ] && timeout 60 ./gpk_selftest --set 53 $mode --batched 0 2>&1 | grep BATCHED | sed "s/^/[knob 53 = $mode] /" >> $O/r06_native_perf_batched.log; done
fi
ls -la $O | head -80
echo "finished at $SECONDS s"
