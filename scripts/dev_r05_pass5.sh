#!/bin/bash
# Round 5, fifth GPU pass: knob sweeps around the factorisation with rows under the matrix (tail panel width, tail length, aggregation
# depth), on the dev library, one box; the bench A/B of the deferred block in the posterior-first order.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd $R/stheno_amd/csrc
{
echo "== default (tail panels 1024, tail 6144, m = 2)"; ./gpk_selftest --perf-rows f64 16384 2048 1024 0 3
echo "== tail panels 512";   ./gpk_selftest --set 52 512 --perf-rows f64 16384 2048 1024 0 3
echo "== tail panels 2048";  ./gpk_selftest --set 52 2048 --perf-rows f64 16384 2048 1024 0 3
echo "== tail 4096";         ./gpk_selftest --set 9 4096 --perf-rows f64 16384 2048 1024 0 3
echo "== tail 5120";         ./gpk_selftest --set 9 5120 --perf-rows f64 16384 2048 1024 0 3
echo "== tail 8192";         ./gpk_selftest --set 9 8192 --perf-rows f64 16384 2048 1024 0 3
echo "== m = 1";             ./gpk_selftest --set 47 1 --perf-rows f64 16384 2048 1024 0 3
echo "== fill workers 1/2 (knob 39 = 128)"; ./gpk_selftest --set 39 128 --perf-rows f64 16384 2048 1024 0 3
echo "== fill workers fewer (knob 39 = 40)"; ./gpk_selftest --set 39 40 --perf-rows f64 16384 2048 1024 0 3
echo "== fp32 N = 32768 default"; ./gpk_selftest --perf-rows f32 32768 2048 1024 512 2
echo "== fp32 tail panels 512"; ./gpk_selftest --set 52 512 --perf-rows f32 32768 2048 1024 512 2
echo "== fp32 tail 4096"; ./gpk_selftest --set 9 4096 --perf-rows f32 32768 2048 1024 512 2
echo "== fp32 tail 8192"; ./gpk_selftest --set 9 8192 --perf-rows f32 32768 2048 1024 512 2
} > $O/perf_rows_sweeps.log 2>&1
grep -E "^==|round [23]" $O/perf_rows_sweeps.log | awk '/^==/{print} /round/{print "   ", $0}' | cut -c1-150
cd /tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], round(d["value"],3), d["unit"], round(d["ms_per_step"],3), "ms", round(r["frac"],4), "kernel ms", round(r["kernel_ms_per_step"],3), "whole", round(d["whole_step"]["frac"],4))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for rep in 1 2; do
  timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched-record 2>/dev/null | grep "^{" | tail -1 > $O/ab_pf_$rep.json; line $O/ab_pf_$rep.json "posterior-first rep$rep"
  timeout 300 python $R/bench.py --steps 20 --warmup 5 --deferred-checks --no-cpu-baseline --no-batched-record 2>/dev/null | grep "^{" | tail -1 > $O/ab_pfd_$rep.json; line $O/ab_pfd_$rep.json "posterior-first deferred rep$rep"
done
echo "finished at $SECONDS s"
