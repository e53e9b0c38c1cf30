#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 500 ./gpk_selftest > $O/selftest_dev.log 2>&1; echo "selftest(dev) rc=$? $(tail -1 $O/selftest_dev.log)"; grep FAIL $O/selftest_dev.log | head
timeout 400 ./gpk_selftest_rel > $O/selftest_rel.log 2>&1; echo "selftest(release) rc=$? $(tail -1 $O/selftest_rel.log)"
cd $R
timeout 400 python scripts/dev_stream_race.py 3000 2>&1 | tail -4 | cut -c1-600 | tee $O/stream_race_after_fix.log
cd $R/stheno_amd/csrc
timeout 120 ./gpk_selftest --perf-pipe 2>&1 | grep -E "pipe=1" | head -24 | tee $O/perf_pipe_after_fix.log
timeout 60 ./gpk_selftest --batched 512 2>&1 | head -3
timeout 100 ./gpk_selftest --perf-rows f64 16384 2048 1024 0 3 2>&1 | tail -4
cd /tmp
timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dense', round(d['ms_per_step'],3), 'ms', round(d['value'],3), 'evals/s; batched', round(d['batched']['ms_per_step'],3))"
echo "finished at $SECONDS s"
