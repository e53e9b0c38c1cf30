#!/bin/bash
# Round 6, the state at HEAD: the whole GPU suite, the four bench lines with this round's PMC files in place, kernel stats of cfg4 (whose kernel changed last)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_final
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -s 2> $O/r06_pytest_gpu.stderr.log | grep -E "ACHIEVED|passed|failed|FAILED|Error" | tee $O/r06_pytest_gpu.log | tail -3
[ -f gpurun_out/r05/achieved_errors.json ] && cp gpurun_out/r05/achieved_errors.json $O/r06_achieved_errors.json
[ -f gpurun_out/r06/achieved.json ] && cp gpurun_out/r06/achieved.json $O/r06_achieved.json
for f in r05_bench_batched_f32_rccl_1rank.json r05_bench_batched_f32_rccl_1rank.stderr.log; do [ -f gpurun_out/r05/$f ] && cp gpurun_out/r05/$f $O/${f/r05_/r06_}; done
cd /tmp
timeout 300 python $R/bench.py --steps 20 --warmup 5 2> $O/r06_final_bench_dense_f64.stderr.log | grep "^{" | tail -1 > $O/r06_final_bench_dense_f64.json
for w in sum_f32 batched_f32 sparse_f32; do
  timeout 300 python $R/bench.py --workload $w --no-batched-record 2> $O/r06_final_bench_$w.stderr.log | grep "^{" | tail -1 > $O/r06_final_bench_$w.json
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b -o s -- python $R/bench.py --workload batched_f32 --steps 5 --warmup 2 --no-cpu-baseline --no-batched-record > $O/r06_stats_batched_f32.log 2>&1
F=$(find $O/stats_b -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/r06_bench_batched_f32_kernel_stats.csv; rm -rf $O/stats_b
timeout 200 python $R/scripts/collect_pmc.py batched_f32 $O/r06_pmc_batched_f32.json > $O/r06_pmc_batched_f32.log 2>&1
timeout 200 python $R/scripts/collect_sq.py batched_f32 $O/r06_sq_batched_f32.json > $O/r06_sq_batched_f32.summary.log 2>&1
python - <<PY
import json
for w in ["dense_f64","sum_f32","batched_f32","sparse_f32"]:
    d=json.load(open("$O/r06_final_bench_%s.json" % w)); r=d["roofline"]
    print(w, round(d["value"],3), round(d["ms_per_step"],3), r["kernel"], round(r["frac"],3), round(r["frac_of_measured"],3), r["traffic"], round(d["whole_step"]["frac"],3), (d.get("batched") or {}).get("ms_per_step"))
PY
echo "finished at $SECONDS s"
