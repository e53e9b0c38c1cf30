#!/bin/bash
# Re-collect ONE workload's round-5 files (bench line, kernel stats, PMC, SQ) after a change that touches only it:
#   W=sparse_f32 bash scripts/collect_r05_one.sh
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05_profiles
mkdir -p $O
w=${W:-sparse_f32}
cd /tmp
timeout 300 python $R/bench.py --workload $w --no-batched-record 2> $O/r05_bench_$w.stderr.log | grep "^{" | tail -1 > $O/r05_bench_$w.json
[ -s $O/r05_bench_$w.json ] || echo "NO BENCH LINE for $w -- see $O/r05_bench_$w.stderr.log"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-batched-record > $O/r05_stats_$w.log 2>&1
F=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/r05_bench_${w}_kernel_stats.csv || echo "NO KERNEL STATS for $w"
rm -rf $O/stats_$w
timeout 200 python $R/scripts/collect_pmc.py $w $O/r05_pmc_$w.json > $O/r05_pmc_$w.log 2>&1 || echo "pmc $w failed"
timeout 200 python $R/scripts/collect_sq.py $w $O/r05_sq_$w.json > $O/r05_sq_$w.summary.log 2>&1 || echo "sq $w failed"
head -3 $O/r05_sq_$w.summary.log | cut -c1-170; head -2 $O/r05_pmc_$w.log
python -c "
import json; d=json.load(open('$O/r05_bench_$w.json')); r=d['roofline']; print('$w', round(d['value'],3), d['unit'], round(d['ms_per_step'],3), 'ms', r['kernel'], round(r['frac'],4), 'whole', round(d['whole_step']['frac'],4))"
