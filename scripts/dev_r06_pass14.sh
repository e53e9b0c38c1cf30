#!/bin/bash
# per-kernel times of ONE batched factorisation at the per-rank batch of an 8-GPU run (64 x 2048)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_pass14
mkdir -p $O
cd /tmp
rm -rf /tmp/tr64
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr64 -o t -- $R/stheno_amd/csrc/gpk_selftest --set 53 1 --batched 0 64 > $O/run.log 2>&1
F=$(find /tmp/tr64 -name "*kernel_trace.csv" | head -1)
python3 - "$F" > $O/steps_batch64.txt <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'kmat' in n]
a,b=idx[2],idx[3]
seq=rows[a+1:b]
def short(n):
    n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'void ','',n); return n.split('(')[0][:50]
t0=int(seq[0]['Start_Timestamp']); prev=t0
tot={}
for r in seq:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{(st-t0)/1e3:9.1f} us  dur {(en-st)/1e3:8.1f}  gap {(st-prev)/1e3:6.1f}  {short(r['Kernel_Name'])}")
    tot[short(r['Kernel_Name'])]=tot.get(short(r['Kernel_Name']),0)+(en-st)/1e3
    prev=en
print('TOTAL span us', (int(seq[-1]['End_Timestamp'])-t0)/1e3)
for k,v in sorted(tot.items(), key=lambda x:-x[1]): print(f"{v:10.1f} us  {k}")
PY
tail -8 $O/steps_batch64.txt
