#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_clean_cache_gpu.py tests/test_learn_gpu.py -x -q -m gpu > gpurun_out/r6c_test.log 2>&1
tail -25 gpurun_out/r6c_test.log
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-utts 0 > gpurun_out/r6c_bench.json 2> gpurun_out/r6c_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6c_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'])
print('epoch_from_files', {k:v for k,v in d.get('epoch_from_files',{}).items() if k in ('value','ms_per_epoch','resident_batches','error')})
print('cached', d.get('epoch_from_files_cached'))
PY
tail -3 gpurun_out/r6c_bench.err
