#!/bin/bash
# Round-3 measurement pass on the GPU box (run from the repo root through gpurun): bench lines, rocprofv3 kernel-trace
# summaries of the same commands, PMC (HBM-side traffic) passes, native self-test + micro-benchmarks, the kernel sequence of
# one dense eval.  Everything lands in gpurun_out/r03_profiles/ (copied to profiles/ afterwards).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_profiles
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 600 ./gpk_selftest > $O/r03_selftest.log 2>&1; echo "selftest rc=$?"; tail -1 $O/r03_selftest.log
timeout 600 ./gpk_selftest --only-perf > $O/r03_native_perf.log 2>&1
timeout 300 ./gpk_selftest --perf-la > $O/r03_native_perf_lookahead.log 2>&1
timeout 200 ./gpk_selftest --perf-kmat > $O/r03_native_perf_kmat.log 2>&1
timeout 300 ./gpk_selftest --perf-trsm > $O/r03_native_perf_trsm.log 2>&1
timeout 200 ./gpk_selftest --perf-pipe > $O/r03_native_perf_pipelined_panel.log 2>&1
timeout 300 ./gpk_selftest --perf-la-tail > $O/r03_native_perf_plain_vs_lookahead.log 2>&1
( cd $R/scripts/dev && timeout 60 ./store_cost ) > $O/r03_store_cost.log 2>&1
timeout 200 ./gpk_selftest --diagprof 2048 > $O/r03_diag_kernel_phases.log 2>&1
( cd $R/scripts/dev && for b in mfma_latency store_cost; do [ -x $b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $b $b.hip; done ) > /dev/null 2>&1   # (git-ignored binaries)
( cd $R/scripts/dev && timeout 60 ./mfma_latency ) > $O/r03_mfma_latency.log 2>&1
cd /tmp
for w in dense_f64 sum_f32 batched_f32 sparse_f32; do
  timeout 400 python $R/bench.py --workload $w 2>/dev/null | grep "^{" | tail -1 > $O/r03_bench_$w.json
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_$w.log 2>&1
  F=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/r03_bench_${w}_kernel_stats.csv
  if [ "$w" = dense_f64 ]; then
    T=$(find $O/stats_$w -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python $R/scripts/dev_trace_sequence.py $T kmat 2 > $O/r03_dense_f64_kernel_sequence.txt 2>&1
  fi
  rm -rf $O/stats_$w
  timeout 600 python $R/scripts/collect_pmc.py $w $O/r03_pmc_$w.json > $O/r03_pmc_$w.log 2>&1
done
GPK_BENCH_FORCE_DIST=1 NCCL_DEBUG=WARN NCCL_DEBUG_FILE=$O/r03_bench_batched_f32_rccl_1rank_nccl.log MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python $R/bench.py --workload batched_f32 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > $O/r03_bench_batched_f32_rccl_1rank.json
ls -la $O | head -50
