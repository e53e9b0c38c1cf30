#!/bin/bash
# The plain tail of the look-ahead WITH rows under the matrix: panel width of the pipelined plain path (knob 52) against the tail's
# length (knob 9), and how the tail's launches split their workgroups between panel tasks and fill tiles (knobs 38 / 39) --
# the sweeps of rounds 4-5 were made without rows (profiles/r06_native_perf_pipelined_panel.log), round 6 re-swept only the length.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass16
mkdir -p $O
cd $R/stheno_amd/csrc
L=$O/tail_panel_width_with_rows.log
: > $L
run() {   # run "<knob settings>" <perf-rows args...>
  local sets="$1"; shift
  timeout 120 ./gpk_selftest $sets --perf-rows "$@" 2>&1 | grep "PERFROWS" | sed "s/^/[$sets] /" >> $L
}
for tail in 4096 6144 8192; do
  for nbo in 512 1024 2048; do
    run "--set 9 $tail --set 52 $nbo" f64 16384 2048 1024 0 3
  done
done
for wgs in 32 64 128 192; do run "--set 39 $wgs" f64 16384 2048 1024 0 3; done
run "--set 38 0" f64 16384 2048 1024 0 3
for nbo in 512 1024 2048; do
  run "--set 52 $nbo" f32 32768 2048 1024 512 2
done
grep "with the rows" $L | awk '{k=$0; sub(/round.*/, "", k); t[k]=(k in t)? (t[k]<$(NF-2)?t[k]:$(NF-2)) : $(NF-2)} END {for (k in t) print t[k], k}' | sort -n > $O/summary.txt
cat $O/summary.txt
echo "finished at $SECONDS s"
