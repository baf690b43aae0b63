import csv,glob,sys
f=sorted(glob.glob('gpurun_out/prof_r01c/*/*_kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-42:],r['Queue_Id']) for r in rows]
ev.sort()
# step boundaries: first stft_band of each group of 3 -> use imcra kernel (1 per step)
im=[i for i,e in enumerate(ev) if ('imcra_band' in e[2])]
# step = from the stft preceding imcra k to the one preceding imcra k+1
k=len(im)-2
def step_start(idx):
    i=idx
    while i>0 and not ('stft_band' in ev[i][2]): i-=1
    while i>0 and ('stft_band' in ev[i-1][2]): i-=1
    return i
a=step_start(im[k]); b=step_start(im[k+1])
t0=ev[a][0]
print('step span ms',(ev[b][0]-t0)/1e6)
qs=sorted(set(e[3] for e in ev[a:b]))
last_end=t0
for s,e,n,q in ev[a:b]:
    dur=(e-s)/1e3
    if dur>=float(sys.argv[1]) if len(sys.argv)>1 else 100:
        print('%8.3f %8.3f  q%s %-44s %8.1f us'%((s-t0)/1e6,(e-t0)/1e6,qs.index(q),n,dur))
