#!/bin/bash
# the four bench lines at HEAD with this round's PMC files in place (roofline.traffic), from /tmp
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_final
mkdir -p $O
cd /tmp
timeout 300 python $R/bench.py --steps 20 --warmup 5 2> $O/r06_final_bench_dense_f64.stderr.log | grep "^{" | tail -1 > $O/r06_final_bench_dense_f64.json
for w in sum_f32 batched_f32 sparse_f32; do
  timeout 300 python $R/bench.py --workload $w --no-batched-record 2> $O/r06_final_bench_$w.stderr.log | grep "^{" | tail -1 > $O/r06_final_bench_$w.json
done
python - <<PY
import json
for w in ["dense_f64","sum_f32","batched_f32","sparse_f32"]:
    d=json.load(open("$O/r06_final_bench_%s.json" % w)); r=d["roofline"]
    print(w, round(d["value"],3), round(d["ms_per_step"],3), r["kernel"], round(r["frac"],3), round(r["frac_of_measured"],3), r["traffic"], round(d["whole_step"]["frac"],3), (d.get("batched") or {}).get("ms_per_step"))
PY
