import csv,glob,sys
pat=sys.argv[1:] or ['eigh','siib']
f=sorted(glob.glob('gpurun_out/prof_r01c/*/*_kernel_stats.csv'))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total per step ms', tot/6/1e6)
for r in rows:
    if any(p in r['Name'] for p in pat) or pat==['all']: print('%-64s calls %4s avg %9.1f us  /step %7.3f ms'%(r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/6/1e6))
