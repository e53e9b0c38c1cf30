#!/bin/bash
# Round 5, third GPU pass: the factorisation with rows under the matrix (native self-test + Python tests), the fp64 Matern kernel
# matrix on the row-band kernel, same-box A/Bs of the bench (call order, deferred checks), SQ counters of the batched workload.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 200 ./gpk_selftest --rows > $O/selftest_rows.log 2>&1; echo "selftest --rows rc=$?"; tail -1 $O/selftest_rows.log; grep FAIL $O/selftest_rows.log | head -30
timeout 500 ./gpk_selftest > $O/selftest_dev.log 2>&1; echo "selftest(dev) rc=$?"; tail -1 $O/selftest_dev.log; grep FAIL $O/selftest_dev.log | head -20
timeout 400 ./gpk_selftest_rel > $O/selftest_rel.log 2>&1; echo "selftest(release) rc=$?"; tail -1 $O/selftest_rel.log; grep FAIL $O/selftest_rel.log | head -20
timeout 200 ./gpk_selftest --perf-kmat > $O/perf_kmat.log 2>&1; grep -i "matern" $O/perf_kmat.log
cd $R
timeout 900 python -m pytest tests/test_round5_rows.py tests/test_round5_evidence.py -q -s 2>&1 | grep -E "ACHIEVED|passed|failed|Error|error|assert|FAILED" | head -60 | tee $O/pytest_round5.log
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_round5_evidence.py --deselect tests/test_round5_rows.py 2>&1 | tail -15 | tee $O/pytest_gpu.log
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
  timeout 300 python $R/bench.py --steps 20 --warmup 5 --order logpdf-first --no-deferred-checks --no-cpu-baseline --no-batched-record 2>/dev/null | grep "^{" | tail -1 > $O/bench_dense_nodefer_$rep.json
  line $O/bench_dense_nodefer_$rep.json "dense logpdf-first no-deferred rep$rep"
done
for ord in posterior-first logpdf-first; do
  timeout 300 python $R/bench.py --workload sum_f32 --order $ord --no-cpu-baseline --no-batched-record 2> $O/bench_sum_$ord.err | grep "^{" | tail -1 > $O/bench_sum_$ord.json
  line $O/bench_sum_$ord.json "sum_f32 $ord"
done
timeout 300 python $R/bench.py --workload sparse_f32 --no-cpu-baseline --no-batched-record 2> $O/bench_sparse.err | grep "^{" | tail -1 > $O/bench_sparse_f32.json
line $O/bench_sparse_f32.json "sparse_f32"
timeout 600 python $R/scripts/collect_sq.py batched_f32 $O/r05_sq_batched_f32.json 2>&1 | tail -8
echo "finished at $SECONDS s"
