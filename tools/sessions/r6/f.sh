#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python tools/learn_curve.py --epochs 30 --noise-tilt 1.5 --out gpurun_out/learn_curve_tilt15.json > gpurun_out/r6f_learn.log 2>&1
grep -v " epoch " gpurun_out/r6f_learn.log | tail -6
grep "f32 epoch" gpurun_out/r6f_learn.log | awk 'NR%3==1' | cut -c1-200
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-utts 0 > gpurun_out/r6f_bench.json 2> gpurun_out/r6f_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6f_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'iff', d.get('inference_from_files',{}).get('value'), 'cached', d.get('epoch_from_files_cached',{}).get('speedup_vs_uncached'))
PY
