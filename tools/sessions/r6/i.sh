#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py --steps 5 --warmup 2 --cpu-utts 0 > gpurun_out/r6i_bench.json 2> gpurun_out/r6i_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6i_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'], 'bound', d.get('host_cpus_bound'))
for k in ('nonperiodic','configs1','shard128','inference','epoch_equivalent','epoch_equivalent_qua','epoch_from_files','epoch_from_files_cached','inference_from_files'):
    v=d.get(k)
    if isinstance(v,dict): print(k, {kk:vv for kk,vv in v.items() if kk in ('value','ms_per_step','ms_per_epoch','speedup_vs_uncached','resident_batches','error','ms_per_step_plain','single_stream','loader_alone')})
PY
tail -3 gpurun_out/r6i_bench.err
