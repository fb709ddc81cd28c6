import csv, sys, bisect
tr=list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r:int(r['Start_Timestamp']))
def grid(r): return (int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']))
lastq={}; out={}
bp=[int(r['Start_Timestamp']) for r in tr if 'bwd_persist<13' in r['Kernel_Name']]
for r in tr:
    q=r['Queue_Id']; s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if q in lastq and 'k_gemm_nt' in r['Kernel_Name'] and grid(r) in [(105,4),(50,6),(40,8)]:
        out.setdefault(grid(r),[]).append((s-lastq[q])/1e3)
    lastq[q]=e
for k,v in out.items():
    v=sorted(v); print(k,'n',len(v),'gap after previous kernel on its queue: median %.1f  min %.1f  max %.1f us'%(v[len(v)//2],v[0],v[-1]))
