import re,sys
lines=[l for l in open(sys.argv[1]) if ' us  dur' in l]
out=[]
for l in lines:
    m=re.match(r'\s*([\d.]+) us  dur\s+([\d.]+)  gap\s+([-\d.]+)  wgs\s+(\d+)  (.*)',l)
    out.append((float(m[1]),float(m[2]),float(m[3]),int(m[4]),m[5].strip()))
acc=None
for t,d,g,w,n in out:
    if d>150:
        if acc: print(f"{acc[0]:9.0f}  [{acc[2]} small kernels, span {acc[1]-acc[0]:7.0f} us, busy {acc[3]:7.0f}]"); acc=None
        print(f"{t:9.0f}  dur {d:7.0f} gap {g:6.1f} wgs {w:6d} {n[:50]}")
    else:
        if not acc: acc=[t,t+d,1,d]
        else: acc[1]=max(acc[1],t+d); acc[2]+=1; acc[3]+=d
if acc: print(f"{acc[0]:9.0f}  [{acc[2]} small kernels, span {acc[1]-acc[0]:7.0f} us, busy {acc[3]:7.0f}]")
