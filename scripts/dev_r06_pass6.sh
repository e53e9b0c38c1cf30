#!/bin/bash
# Round 6, pass 6: the mixed-phase batched steps as shipped (one task per workgroup, placement check, 128-register diagonal blocks for large fp32 batches):
# full self-tests of both libraries, the batched GPU tests, A/B timings, the batched bench line.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass6
mkdir -p $O
cd $R/stheno_amd/csrc
timeout 500 ./gpk_selftest > $O/selftest.log 2>&1; echo "selftest(dev) rc=$? $(tail -1 $O/selftest.log)"
timeout 400 ./gpk_selftest_rel > $O/selftest_release.log 2>&1; echo "selftest(release) rc=$? $(tail -1 $O/selftest_release.log)"
grep FAIL $O/selftest*.log | head
for mode in "53 0 --set 57 0" "53 0 --set 57 1" "53 1 --set 57 0" "53 1 --set 57 1"; do
  timeout 120 ./gpk_selftest --set $mode --batched 0 2>&1 | grep "BATCHED potrf\|differing" | sed "s/^/[$mode] /" | tee -a $O/batched_ab.log
done
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_round5_concurrency.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -5
cd /tmp
timeout 300 python $R/bench.py --workload batched_f32 --no-cpu-baseline 2> $O/bench_batched.stderr.log | grep "^{" | tail -1 > $O/bench_batched_f32.json
python -c "
import json; d=json.load(open('$O/bench_batched_f32.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['whole_step'])"
echo "finished at $SECONDS s"
