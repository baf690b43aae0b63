"""Concurrency profile of one canonical step from a rocprofv3 kernel trace: python tools/timeline_summary.py <kernel_trace.csv> [bin_ms=1.0]
Takes the last complete step (between two stft_band_wave_kernel launches on the main stream's queue) and prints, per time bin, how
many kernels were running on average, the share of the bin with NO kernel running, and the kernels that held most of it."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
binms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-42:], r['Queue_Id']) for r in rows)
# step boundaries: the first imcra_band_kernel of each step
marks = [s for s, e, n, q in ks if n.endswith('imcra_band_kernel<false>') or 'imcra_band_kernel' in n]
if len(marks) < 3:
    sys.exit('not enough steps in the trace')
t0, t1 = marks[-3], marks[-2]
sel = [(max(s, t0), min(e, t1), n, q) for s, e, n, q in ks if e > t0 and s < t1]
L = (t1 - t0) / 1e6
print('step length %.2f ms, %d launches, kernel time %.2f ms' % (L, len(sel), sum(e - s for s, e, _, _ in sel) / 1e6))
nb = int(L / binms) + 1
busy = [0.0] * nb; idle = [0.0] * nb; names = [collections.Counter() for _ in range(nb)]
ev = []
for s, e, n, q in sel:
    ev.append((s, 1)); ev.append((e, -1))
    b0, b1 = int((s - t0) / 1e6 / binms), int((e - t0) / 1e6 / binms)
    for b in range(b0, min(b1, nb - 1) + 1):
        lo, hi = t0 + b * binms * 1e6, t0 + (b + 1) * binms * 1e6
        ov = max(0.0, min(e, hi) - max(s, lo))
        busy[b] += ov; names[b][n] += ov
ev.sort(); cur = 0; last = t0
for t, d in ev:
    if cur == 0 and t > last:
        b = int((last - t0) / 1e6 / binms)
        idle[min(b, nb - 1)] += (t - last)
    cur += d; last = max(last, t) if cur == 0 else last
    if cur == 0: last = t
tot_idle = sum(idle) / 1e6
print('no kernel running: %.2f ms of the step' % tot_idle)
for b in range(nb):
    w = binms * 1e6
    top = ', '.join('%s %.0f%%' % (n, 100 * v / w) for n, v in names[b].most_common(3))
    print('%5.1f ms  concurrency %.2f  idle %3.0f%%  %s' % (b * binms, busy[b] / w, 100 * idle[b] / w, top))
