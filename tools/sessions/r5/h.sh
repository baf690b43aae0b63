#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "plan" 2>&1 | tail -8
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tools/host_time.py 32 2>&1 | tail -6
timeout 900 python bench.py --cpu-utts 0 > gpurun_out/r5h_bench.json 2> gpurun_out/r5h_bench.err; tail -3 gpurun_out/r5h_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r5h_bench.json').read().strip().splitlines()[-1])
print('headline', round(d['ms_per_step'],2), 'utt/s', round(d['value']), d.get('step_status'))
for k in ('configs1','nonperiodic','headline_f32','headline_pipelined','shard128','global1024','inference','epoch_equivalent'):
    v=d.get(k,{})
    print(k, {q: (round(v[q],3) if isinstance(v[q],float) else v[q]) for q in v if q in ('ms_per_step','ms_per_step_plain','value','ms_per_batch','ms_per_unit','error','imcra_ms','by_batch','single_stream')})
P
