#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp
for ord in posterior-first logpdf-first; do
  timeout 200 python $R/scripts/dev_host_profile.py $ord > $O/host_profile_$ord.log 2>&1
  echo "== $ord"; head -60 $O/host_profile_$ord.log
done
echo "finished at $SECONDS s"
