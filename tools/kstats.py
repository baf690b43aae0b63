"""Per-kernel table of the newest rocprofv3 kernel_stats.csv under a directory: python tools/kstats.py gpurun_out/r03_prof/stats_serial [steps] [rows]"""
import csv, glob, os, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0; nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = max(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f, ' total ms/step %.2f' % (tot / steps / 1e6))
fam = {'haspi': 0, 'eigh': 0, 'siib': 0, 'conv16': 0, 'wgrad': 0, 'estoi': 0, 'other': 0}
for r in rows:
    k = next((key for key in fam if key in r['Name']), 'other')
    fam[k] += float(r['TotalDurationNs']) / steps / 1e6
print({k: round(v, 2) for k, v in fam.items()})
for r in rows[:nrows]:
    print('%-84s %5s %8.3f ms/step %8.1f us' % (r['Name'][:84], r['Calls'], float(r['TotalDurationNs']) / steps / 1e6, float(r['AverageNs']) / 1e3))
