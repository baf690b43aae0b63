#!/bin/bash
# round 5, call d: whole GPU suite on the fused generator path + bench line
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py --cpu-utts 0 > gpurun_out/r5d_bench.json 2> gpurun_out/r5d_bench.err; tail -3 gpurun_out/r5d_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r5d_bench.json').read().strip().splitlines()[-1])
print('headline', round(d['ms_per_step'],2), 'utt/s', round(d['value']))
for k in ('configs1','nonperiodic','headline_f32','headline_pipelined','shard128','global1024','inference','epoch_equivalent'):
    v=d.get(k,{})
    print(k, {q: (round(v[q],3) if isinstance(v[q],float) else v[q]) for q in v if q in ('ms_per_step','ms_per_step_plain','value','ms_per_batch','ms_per_unit','error')})
P
