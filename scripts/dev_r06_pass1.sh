#!/bin/bash
# Round 6, first pass on the GPU box: measured MFMA peak, the round-5 ADVICE regressions on HIP, baseline bench lines with a clock / power trace.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass1
mkdir -p $O
timeout 600 python -m pytest tests/test_round6_evidence.py tests/test_round5_host.py tests/test_round5_rows.py -m gpu -q -s -x 2> $O/pytest.stderr.log | grep -E "ACHIEVED|passed|failed|FAILED|Error" | tee $O/pytest.log | tail -12
cp -r gpurun_out/r06 $O/ 2>/dev/null
cd /tmp
timeout 300 python $R/scripts/clock_trace.py $O/clock_trace_dense_f64.json -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-batched-record 2> $O/clock_trace.stderr.log | cut -c1-1500
timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_dense.stderr.log | grep "^{" | tail -1 > $O/bench_dense_f64.json
python -c "
import json; d=json.load(open('$O/bench_dense_f64.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('peak_measured'), d['roofline'].get('frac_of_measured'), d['whole_step']['frac']); print(d['batched']['ms_per_step'], d['batched']['per_gpu_step'])"
amd-smi metric -g 0 --clock --power 2>&1 | head -40 > $O/amd_smi_metric.txt
echo "finished at $SECONDS s"
