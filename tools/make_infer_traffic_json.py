"""Summarise the PMC passes of tools/prof_infer.sh into profiles/<round>/infer_traffic.json (every kernel of the enhancement path).

usage: python tools/make_infer_traffic_json.py gpurun_out/r05_infer [round=r05]
Same corrections as tools/make_traffic_json.py: HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE (KB counters, gfx950), MFMA-busy =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz); per-launch averages of the newest run in each pass directory.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1]
ROUND = sys.argv[2] if len(sys.argv) > 2 else 'r05'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def short(name):
    m = re.match(r'(?:void )?([A-Za-z_0-9]+(?:<[^(]*>)?)', name)
    return m.group(1) if m else name


acc = defaultdict(lambda: defaultdict(list))
for d in ('pmc_FETCH_SIZE', 'pmc_WRITE_SIZE', 'pmc_SQ'):
    for f in sorted(glob.glob('%s/%s/*/*_counter_collection.csv' % (root, d)), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k.startswith('at::') or 'elementwise' in k or k.startswith('__amd'):
                continue
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if d == 'pmc_SQ' and r['Counter_Name'] == 'SQ_WAVE_CYCLES':
                acc[k]['_dur_ns'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
out = {'note': 'PMC passes of tools/prof_infer.sh (python tools/infer_time.py 128 10 plain: Enhancer.enhance on 128 x 8 s utterances, single stream '
               'so that every kernel runs alone; separate rocprofv3 passes for FETCH_SIZE, WRITE_SIZE and the SQ counters).  HBM bytes per launch = '
               'FETCH_SIZE x 2 + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md); MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz).  '
               'Per-launch averages.',
       'csrc_sha': bench.csrc_sha(), 'kernels': {}}
for k, c in sorted(acc.items()):
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    e = {'launches': len(c.get('FETCH_SIZE', c.get('SQ_WAVE_CYCLES', [])))}
    if 'FETCH_SIZE' in avg and 'WRITE_SIZE' in avg:
        e['hbm_bytes_corrected'] = int(avg['FETCH_SIZE'] * 1024 * 2 + avg['WRITE_SIZE'] * 1024)
    if avg.get('_dur_ns'):
        e['pmc_pass_duration_us'] = round(avg['_dur_ns'] / 1e3, 1)
        e['mfma_busy_frac'] = round(avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024 * avg['_dur_ns'] * 2.4), 3)
    out['kernels'][k] = e
json.dump(out, open('profiles/%s/infer_traffic.json' % ROUND, 'w'), indent=1)
for k, e in out['kernels'].items():
    print('%-46s %s' % (k[:46], e))
