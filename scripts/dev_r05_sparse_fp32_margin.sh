#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== round 5 math"; python scripts/dev_r05_sparse_fp32_margin.py 2>&1 | tail -9
cp stheno_amd/csrc/ab_r4math/libgpk.so stheno_amd/csrc/dev/libgpk.so
echo "== round 4 math"; GPK_DEV=1 python scripts/dev_r05_sparse_fp32_margin.py 2>&1 | tail -9
