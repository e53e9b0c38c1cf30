#!/bin/bash
# Round-2 measurement pass on the GPU box (run from the repo root through gpurun): bench lines, rocprofv3
# kernel-trace summaries of the same commands, PMC (HBM-side traffic) passes, native self-test + micro-benchmarks,
# and the look-ahead overlap trace.  Everything lands in gpurun_out/r02_profiles/ (copied to profiles/ afterwards).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_profiles
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 600 ./gpk_selftest > $O/r02_selftest.log 2>&1; echo "selftest rc=$?"; tail -1 $O/r02_selftest.log
timeout 600 ./gpk_selftest --only-perf > $O/r02_native_perf.log 2>&1
timeout 300 ./gpk_selftest --perf-la > $O/r02_native_perf_lookahead.log 2>&1
timeout 200 ./gpk_selftest --perf-kmat > $O/r02_native_perf_kmat.log 2>&1
( timeout 100 ./gpk_selftest --tileprof 1024 1 0; timeout 100 ./gpk_selftest --tileprof 1024 1 8; timeout 100 ./gpk_selftest --tileprof 8192 1 0 ) > $O/r02_tile_profile.log 2>&1
( timeout 200 ./gpk_selftest --la-clock f64 16384 1024 0; timeout 200 ./gpk_selftest --la-clock f64 16384 1024 10 ) > $O/r02_la_clock.log 2>&1
cd /tmp
for w in dense_f64 sum_f32 batched_f32 sparse_f32; do
  timeout 400 python $R/bench.py --workload $w 2>/dev/null | tail -1 > $O/r02_bench_$w.json
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_$w.log 2>&1
  F=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/r02_bench_${w}_kernel_stats.csv
  rm -rf $O/stats_$w
  timeout 600 python $R/scripts/collect_pmc.py $w $O/r02_pmc_$w.json > $O/r02_pmc_$w.log 2>&1
done
rm -f $O/r02_sq_counters.txt
for shape in "15360 15360 1024 1" "15360 15360 4096 1" "8192 8192 8192 0"; do
  timeout 250 python $R/scripts/dev_sq_counters.py $O/r02_sq_counters.txt -- $R/stheno_amd/csrc/gpk_selftest --gemm f64 $shape > /dev/null 2>&1
done
# the look-ahead factorisation alone: kernel trace (two queues) + overlap summary
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $R/stheno_amd/csrc/gpk_selftest --la-one f64 16384 1024 1 2048 2 > $O/trace.log 2>&1
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$F" ]; then cp $F $O/r02_potrf_lookahead_f64_n16384_kernel_trace.csv; python $R/scripts/dev_trace_overlap.py $F > $O/r02_potrf_lookahead_f64_n16384_overlap.txt 2>&1; fi
rm -rf $O/trace
ls -la $O | head -40
