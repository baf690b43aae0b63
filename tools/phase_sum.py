import csv,glob,sys,collections
f=sorted(glob.glob('gpurun_out/prof_r01c/*/*_kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-40:],r['Queue_Id']) for r in rows]
ev.sort()
im=[i for i,e in enumerate(ev) if ('imcra_band' in e[2])]
k=len(im)-2
def step_start(idx):
    i=idx
    while i>0 and not ('stft_band' in ev[i][2]): i-=1
    while i>0 and ('stft_band' in ev[i-1][2]): i-=1
    return i
a=step_start(im[k]); b=step_start(im[k+1])
t0=ev[a][0]
qs=sorted(set(e[3] for e in ev[a:b]))
lo=float(sys.argv[1]); hi=float(sys.argv[2])
acc=collections.defaultdict(lambda:[0,0.0])
for s,e,nm,q in ev[a:b]:
    if qs.index(q)!=0: continue
    ts=(s-t0)/1e6
    if ts<lo or ts>hi: continue
    acc[nm][0]+=1; acc[nm][1]+=(e-s)/1e3
tot=0
for nm,(n,t) in sorted(acc.items(), key=lambda x:-x[1][1]):
    print('%-42s n=%3d  %8.1f us'%(nm,n,t)); tot+=t
print('total %.1f us'%tot)
