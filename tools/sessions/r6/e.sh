#!/bin/bash
# r6 call e: the whole -m gpu suite + the default bench line
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r6e_test.log 2>&1
tail -15 gpurun_out/r6e_test.log
timeout 900 python bench.py > gpurun_out/r6e_bench.json 2> gpurun_out/r6e_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6e_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'])
for k in ('nonperiodic','configs1','shard128','inference','epoch_equivalent','epoch_equivalent_qua','epoch_from_files','epoch_from_files_cached','inference_from_files','cpu_baseline'):
    v=d.get(k)
    if isinstance(v,dict): print(k, {kk:vv for kk,vv in v.items() if kk in ('value','ms_per_step','ms_per_epoch','speedup_vs_uncached','resident_batches','error','ms_per_step_plain','single_stream')})
PY
tail -3 gpurun_out/r6e_bench.err
