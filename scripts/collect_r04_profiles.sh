#!/bin/bash
# Round-4 measurement pass on the GPU box (run from the repo root through gpurun): bench lines, rocprofv3 kernel-trace
# summaries of the same commands, PMC (HBM-side traffic) passes, native self-test + micro-benchmarks, the kernel sequence of
# one dense eval, the one-off full-size CPU baseline.  Everything lands in gpurun_out/r04_profiles/ (copied to profiles/ afterwards).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04_profiles
mkdir -p $O
# WORKLOADS="sparse_f32" bash scripts/collect_r04_profiles.sh  re-collects one workload's files only
if [ -z "$WORKLOADS" ]; then
cd $R/stheno_amd/csrc
timeout 600 ./gpk_selftest > $O/r04_selftest.log 2>&1; echo "selftest rc=$?"; tail -1 $O/r04_selftest.log
timeout 600 ./gpk_selftest --only-perf > $O/r04_native_perf.log 2>&1
timeout 300 ./gpk_selftest --perf-la > $O/r04_native_perf_lookahead.log 2>&1
timeout 300 ./gpk_selftest --perf-trsm > $O/r04_native_perf_trsm.log 2>&1
fi
cd /tmp
for w in ${WORKLOADS:-dense_f64 sum_f32 batched_f32 sparse_f32}; do
  if [ "$w" = dense_f64 ]; then
    timeout 400 python $R/bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" | tail -1 > $O/r04_bench_$w.json
  else
    timeout 400 python $R/bench.py --workload $w --no-batched-record 2>/dev/null | grep "^{" | tail -1 > $O/r04_bench_$w.json
  fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-batched-record > $O/stats_$w.log 2>&1
  F=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/r04_bench_${w}_kernel_stats.csv
  if [ "$w" = dense_f64 ]; then
    T=$(find $O/stats_$w -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python $R/scripts/dev_trace_sequence.py $T kmat 2 > $O/r04_dense_f64_kernel_sequence.txt 2>&1
  fi
  rm -rf $O/stats_$w $O/stats_$w.log
  timeout 600 python $R/scripts/collect_pmc.py $w $O/r04_pmc_$w.json > $O/r04_pmc_$w.log 2>&1
done
[ -n "$WORKLOADS" ] && exit 0
GPK_BENCH_FORCE_DIST=1 NCCL_DEBUG=WARN MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python $R/bench.py --gpus 1 --workload batched_f32 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > $O/r04_bench_batched_f32_rccl_1rank.json
timeout 400 python $R/bench.py --cpu-baseline-full 2>/dev/null | grep "^{" | tail -1 > $O/r04_cpu_baseline_full_dense_f64.json
ls -la $O | head -50
