"""Timeline of the last step in a rocprofv3 kernel trace: per queue, kernels longer than a threshold with start/end in ms from the
step's first kernel.  usage: python tools/trace_timeline.py <kernel_trace.csv> [min_us] [step_marker_kernel]"""
import csv, sys
f = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
marker = sys.argv[3] if len(sys.argv) > 3 else 'imcra_band_kernel'
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [int(r['Start_Timestamp']) for r in rows if marker in r['Kernel_Name']]
# last full step: between the last two markers
t0, t1 = starts[-2], starts[-1]
sel = [r for r in rows if t0 - 2_000_000 <= int(r['Start_Timestamp']) < t1]
qs = {}
for r in sel:
    qs.setdefault(r['Queue_Id'], []).append(r)
print('step span %.2f ms, %d kernels' % ((t1 - t0) / 1e6, len(sel)))
for q, lst in sorted(qs.items()):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in lst) / 1e6
    print('--- queue %s: %d kernels, busy %.2f ms' % (q, len(lst), busy))
    for r in lst:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if d >= min_us:
            print('  %8.2f - %8.2f  %8.1f us  %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6, d, r['Kernel_Name'][:70]))
