#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "adam" 2>&1 | tail -3
timeout 1200 python bench.py --cpu-utts 0 > gpurun_out/r5k_bench.json 2> gpurun_out/r5k_bench.err; tail -3 gpurun_out/r5k_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r5k_bench.json').read().strip().splitlines()[-1])
print('headline', round(d['ms_per_step'],2), 'utt/s', round(d['value']))
for k in ('headline_one_stream','headline_pipelined','epoch_equivalent','epoch_equivalent_qua','epoch_from_files'):
    print(k, json.dumps(d.get(k))[:700])
P
