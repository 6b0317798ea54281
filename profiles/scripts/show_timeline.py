#!/usr/bin/env python3
"""Print the kernels between the last two marker fills of a rocprofv3 kernel trace (see eager_steady.py)."""
import csv,re,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'FillFunctor<float>' in r['Kernel_Name'] and int(r['Grid_Size_X'])<=256 and int(r['Workgroup_Size_X'])<=256]
# last two marker fills (each marker = zeros + fill -> take the 'fill_' ones: last of pair)
a,b=marks[-3],marks[-1]
seg=[r for r in rows[a+1:b] if not ('FillFunctor<float>' in r['Kernel_Name'] and int(r['Grid_Size_X'])<=256)]
t0=int(seg[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in seg)
print("kernels",len(seg),"span us",(t1-t0)/1e3, "sum us", sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/1e3)
# idle time: union of intervals
iv=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in seg)
busy=0; cs,ce=iv[0]
for s_,e_ in iv[1:]:
    if s_>ce: busy+=ce-cs; cs,ce=s_,e_
    else: ce=max(ce,e_)
busy+=ce-cs
print("busy us", busy/1e3, "idle us", (t1-t0-busy)/1e3)
for r in seg:
    n=r['Kernel_Name']; n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'^void ','',n)
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} q{r.get('Queue_Id','?')} g{r['Grid_Size_X']:>8s} {n[:60]}")
