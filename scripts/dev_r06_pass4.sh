#!/bin/bash
# Round 6, pass 4: per-kernel times of one batched factorisation, lockstep launches against mixed-phase steps (rocprofv3 kernel trace).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass4
mkdir -p $O
cd /tmp
for mode in 0 1; do
  rm -rf /tmp/tr$mode
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$mode -o t -- $R/stheno_amd/csrc/gpk_selftest --set 53 $mode --set 55 256 --batched 0 > $O/run$mode.log 2>&1
  F=$(find /tmp/tr$mode -name "*kernel_trace.csv" | head -1)
  python3 - "$F" $mode > $O/steps_mode$mode.txt <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the LAST gpk_potrf call of the timing loop: find the last kmat kernel, take what follows until the next kmat
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'kmat' in n]
# timing loop: reps 0..2 -> kmat indices 0..2; rep 2 = idx[2] .. idx[3]
a,b=idx[2],idx[3]
seq=rows[a+1:b]
def short(n):
    n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'void ','',n); return n.split('(')[0][:60]
t0=int(seq[0]['Start_Timestamp'])
tot={}
for r in seq:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} us  {d:9.1f} us  grid={r.get('Grid_Size','?'):>8}  {short(r['Kernel_Name'])}")
    tot[short(r['Kernel_Name'])]=tot.get(short(r['Kernel_Name']),0)+d
print('TOTAL span us', (int(seq[-1]['End_Timestamp'])-t0)/1e3)
for k,v in sorted(tot.items(), key=lambda x:-x[1]): print(f"{v:10.1f} us  {k}")
PY
  tail -12 $O/steps_mode$mode.txt
done
echo "finished at $SECONDS s"
