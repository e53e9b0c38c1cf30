#!/bin/bash
# Round 5, fourth GPU pass: where does a cfg2 eval spend its time with the posterior's rows in the factorisation, against the
# separate solve -- kernel traces of both call orders on one box; bench A/B of the two orders (info read where the library puts it).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_round5_rows.py tests/test_round5_evidence.py -q -s 2>&1 | grep -E "ACHIEVED|passed|failed|Error|error|assert|FAILED" | head -40 | tee $O/pytest_round5.log
cd /tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], round(d["value"],3), d["unit"], round(d["ms_per_step"],3), "ms", r["kernel"], round(r["frac"],4), "kernel ms", round(r["kernel_ms_per_step"],3), "whole", round(d["whole_step"]["frac"],4))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for rep in 1 2; do
  for ord in posterior-first logpdf-first; do
    timeout 300 python $R/bench.py --steps 20 --warmup 5 --order $ord --no-cpu-baseline --no-batched-record 2> $O/bench_dense_$ord.err | grep "^{" | tail -1 > $O/bench_dense_${ord}_$rep.json
    line $O/bench_dense_${ord}_$rep.json "dense $ord rep$rep"
  done
done
for ord in posterior-first logpdf-first; do
  timeout 300 python $R/bench.py --workload sum_f32 --order $ord --no-cpu-baseline --no-batched-record 2> $O/bench_sum_$ord.err | grep "^{" | tail -1 > $O/bench_sum_$ord.json
  line $O/bench_sum_$ord.json "sum_f32 $ord"
done
for ord in posterior-first logpdf-first; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$ord -o s -- python $R/bench.py --order $ord --steps 5 --warmup 2 --no-cpu-baseline --no-batched-record > $O/stats_$ord.log 2>&1
  F=$(find $O/stats_$ord -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/r05_dense_${ord}_kernel_stats.csv
  T=$(find $O/stats_$ord -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python $R/scripts/dev_trace_sequence.py $T kmat 2 > $O/r05_dense_${ord}_kernel_sequence.txt 2>&1
  rm -rf $O/stats_$ord
  echo "== $ord: totals per kernel (one eval)"; grep -A40 -i "totals" $O/r05_dense_${ord}_kernel_sequence.txt | head -45
done
echo "finished at $SECONDS s"
