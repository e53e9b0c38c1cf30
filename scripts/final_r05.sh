#!/bin/bash
# Round 5, last pass at HEAD: the whole GPU test suite, both self-test binaries, the four bench lines (no profiler).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05_final
mkdir -p $O
( cd $R && timeout 900 python -m pytest tests -q -m gpu --tb=short -x 2> $O/r05_pytest_gpu.stderr.log | tail -15 > $O/r05_pytest_gpu.log )
tail -3 $O/r05_pytest_gpu.log
( cd $R/stheno_amd/csrc && timeout 600 ./gpk_selftest > $O/r05_selftest.log 2>&1; timeout 600 ./gpk_selftest_rel > $O/r05_selftest_release.log 2>&1 )
grep -h "SUMMARY\|FAIL" $O/r05_selftest.log $O/r05_selftest_release.log | head
cd /tmp
for w in dense_f64 sum_f32 batched_f32 sparse_f32; do
  timeout 300 python $R/bench.py --workload $w --no-batched-record 2> $O/r05_bench_$w.stderr.log | grep "^{" | tail -1 > $O/r05_bench_$w.json
  python -c "
import json; d=json.load(open('$O/r05_bench_$w.json')); r=d['roofline']; print('$w', round(d['value'],3), d['unit'], round(d['ms_per_step'],3), 'ms', r['kernel'][:40], round(r['frac'],4), 'whole', round(d['whole_step']['frac'],4))"
done
