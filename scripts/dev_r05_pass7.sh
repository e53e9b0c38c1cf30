#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 300 ./gpk_selftest --rows > $O/selftest_rows.log 2>&1; echo "selftest --rows rc=$? $(tail -1 $O/selftest_rows.log)"; grep FAIL $O/selftest_rows.log | head -20
timeout 200 ./gpk_selftest --batched 512 2>&1 | tee $O/perf_batched_fused.log | tail -8
cd $R
timeout 600 python -m pytest tests/test_round5_rows.py -q -k "batched" 2>&1 | tail -5
cd /tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], round(d["value"],1), d["unit"], round(d["ms_per_step"],3), "ms", r["kernel"], round(r["frac"],4), "whole", round(d["whole_step"]["frac"],4))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for rep in 1 2; do
  timeout 200 python $R/bench.py --workload batched_f32 --no-cpu-baseline 2> $O/bench_batched.err | grep "^{" | tail -1 > $O/ab_batched_fused_$rep.json; line $O/ab_batched_fused_$rep.json "batched (release: fused) rep$rep"
  GPK_DEV=1 GPK_TUNE=53=0 timeout 200 python $R/bench.py --workload batched_f32 --no-cpu-baseline 2>> $O/bench_batched.err | grep "^{" | tail -1 > $O/ab_batched_unfused_$rep.json; line $O/ab_batched_unfused_$rep.json "batched (dev, knob 53 = 0: kmat launch + potrf) rep$rep"
  GPK_DEV=1 timeout 200 python $R/bench.py --workload batched_f32 --no-cpu-baseline 2>> $O/bench_batched.err | grep "^{" | tail -1 > $O/ab_batched_devfused_$rep.json; line $O/ab_batched_devfused_$rep.json "batched (dev, fused) rep$rep"
done
echo "finished at $SECONDS s"
