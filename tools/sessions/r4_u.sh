#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4u; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pu
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pu -- python3 $GRAFT_REPO_ROOT/tools/chains_pipe.py 32 early > /tmp/pu.log 2>&1
tail -4 /tmp/pu.log | cut -c1-400
f=$(ls /tmp/pu/*/*kernel_trace.csv | head -1); [ -z "$f" ] && exit 1; head -1 "$f" | cut -c1-300
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r.get('Stream_Id', ''), r['Kernel_Name'][:50]) for r in rows]
ks.sort()
# last 16 ms of the run: print every kernel with queue and times relative
t_end = ks[-1][1]
sel = [k for k in ks if k[0] > t_end - 30_000_000 and k[0] < t_end - 14_000_000]
t0 = sel[0][0]
out = open(sys.argv[1].rsplit('/', 1)[0] + '/../tl.txt', 'w')
for s, e, q, st, n in sel:
    out.write('%9.3f %9.3f q%s s%s %s\n' % ((s - t0) / 1e6, (e - s) / 1e6, q, st, n))
out.close()
PY
cp /tmp/pu/tl.txt $GRAFT_REPO_ROOT/gpurun_out/r4u/tl.txt 2>/dev/null || cp /tmp/pu/*/../tl.txt $GRAFT_REPO_ROOT/gpurun_out/r4u/tl.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/r4u/tl.txt
