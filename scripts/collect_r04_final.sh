#!/bin/bash
# Round-4 final pass on the GPU box (from the repo root through gpurun): self-test, the ragged GEMM shapes, the whole GPU test suite and,
# if those are green, the bench / rocprofv3 / PMC files of the four workloads -- each step only while the box-time budget allows.
# usage: BUDGET=900 bash scripts/collect_r04_final.sh        (seconds of run time this call may use)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04_profiles
mkdir -p $O
BUDGET=${BUDGET:-900}
left() { echo $((BUDGET - SECONDS)); }
cd $R/stheno_amd/csrc
timeout 300 ./gpk_selftest > $O/r04_selftest.log 2>&1; tail -1 $O/r04_selftest.log
grep -q "fail=0" $O/r04_selftest.log || { echo "SELFTEST FAILED"; grep FAIL $O/r04_selftest.log | head -20; exit 1; }
{
./gpk_selftest --gemm f64 15000 15000 1000 1 | tail -1
./gpk_selftest --gemm f64 8000 2000 15008 0 | tail -1
./gpk_selftest --gemm f64 8000 2000 15000 0 | tail -1
./gpk_selftest --gemm f32 8000 2000 15000 0 | tail -1
./gpk_selftest --gemm f64 8192 2048 15008 0 | tail -1
./gpk_selftest --batched 512 | tail -2
} 2>&1 | tee $O/r04_ragged_gemm.log
cd $R
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/r04_pytest_gpu.log
grep -q " passed" $O/r04_pytest_gpu.log && ! grep -q "failed" $O/r04_pytest_gpu.log || { echo "PYTEST FAILED"; exit 1; }
echo "tests done at $SECONDS s"
cd /tmp
for w in batched_f32 dense_f64 sum_f32 sparse_f32; do
  [ $(left) -lt 150 ] && { echo "skipping $w: $(left) s left"; continue; }
  if [ "$w" = dense_f64 ]; then
    timeout 200 python $R/bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" | tail -1 > $O/r04_bench_$w.json
  else
    timeout 200 python $R/bench.py --workload $w --no-batched-record 2>/dev/null | grep "^{" | tail -1 > $O/r04_bench_$w.json
  fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-batched-record > $O/stats_$w.log 2>&1
  F=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/r04_bench_${w}_kernel_stats.csv
  if [ "$w" = dense_f64 ]; then
    T=$(find $O/stats_$w -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python $R/scripts/dev_trace_sequence.py $T kmat 2 > $O/r04_dense_f64_kernel_sequence.txt 2>&1
  fi
  rm -rf $O/stats_$w $O/stats_$w.log
  echo "$w bench + stats done at $SECONDS s"
done
for w in batched_f32 dense_f64 sum_f32 sparse_f32; do
  [ $(left) -lt 190 ] && { echo "skipping pmc $w: $(left) s left"; continue; }
  timeout 180 python $R/scripts/collect_pmc.py $w $O/r04_pmc_$w.json > $O/r04_pmc_$w.log 2>&1
  echo "$w pmc done at $SECONDS s"
done
cd $R/stheno_amd/csrc
[ $(left) -gt 60 ] && timeout 50 ./gpk_selftest --perf-trsm > $O/r04_native_perf_trsm.log 2>&1
[ $(left) -gt 80 ] && timeout 70 ./gpk_selftest --perf-la > $O/r04_native_perf_lookahead.log 2>&1
cd /tmp
[ $(left) -gt 60 ] && GPK_BENCH_FORCE_DIST=1 NCCL_DEBUG=WARN MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 50 python $R/bench.py --gpus 1 --workload batched_f32 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > $O/r04_bench_batched_f32_rccl_1rank.json
echo "finished at $SECONDS s"
