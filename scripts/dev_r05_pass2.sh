#!/bin/bash
# Round 5, second GPU pass: the single-launch TRSV sweep, the transposed cross-covariance of the pseudo-point path, the round-5 evidence
# tests, the whole GPU suite, bench lines of all four workloads, a first SQ-counter pass.  stderr kept everywhere.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 400 ./gpk_selftest > $O/selftest_dev.log 2>&1; echo "selftest(dev) rc=$?"; tail -1 $O/selftest_dev.log; grep FAIL $O/selftest_dev.log | head -20
timeout 300 ./gpk_selftest_rel > $O/selftest_rel.log 2>&1; echo "selftest(release) rc=$?"; tail -1 $O/selftest_rel.log; grep FAIL $O/selftest_rel.log | head -20
timeout 200 ./gpk_selftest --perf-trsv > $O/perf_trsv.log 2>&1; cat $O/perf_trsv.log | grep -v "^PERFTRSV.*per-block.*ms" | head -40; grep "per-block" $O/perf_trsv.log | awk 'NR%3==0' 
cd $R
timeout 900 python -m pytest tests/test_round5_evidence.py -x -q -s 2>&1 | grep -E "ACHIEVED|passed|failed|Error|error|assert" | head -60 | tee $O/pytest_round5.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_round5_evidence.py 2>&1 | tail -15 | tee $O/pytest_gpu.log
cd /tmp
for w in dense_f64 sum_f32 sparse_f32 batched_f32; do
  if [ "$w" = dense_f64 ]; then
    timeout 300 python $R/bench.py --steps 20 --warmup 5 2> $O/bench_$w.err | grep "^{" | tail -1 > $O/bench_$w.json
  else
    timeout 300 python $R/bench.py --workload $w --no-batched-record --no-cpu-baseline 2> $O/bench_$w.err | grep "^{" | tail -1 > $O/bench_$w.json
  fi
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$w.json")); r=d["roofline"]
    print("$w", round(d["value"],3), d["unit"], round(d["ms_per_step"],3), "ms", r["kernel"], round(r["frac"],4), "whole", round(d["whole_step"]["frac"],4))
except Exception as e:
    print("$w: no line", e)
PY
done
timeout 600 python $R/scripts/collect_sq.py dense_f64 $O/r05_sq_dense_f64.json 2>&1 | tail -8
echo "finished at $SECONDS s"
