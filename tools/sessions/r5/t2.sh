#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_varlen_gpu.py tests/test_step_parity_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -8
for k in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --companions 0 --cpu-utts 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_mfma']
print('%.2f ms/step  conv5 fwd in-step %.3f ms, isolated %.3f ms (frac %.3f)' % (d['ms_per_step'], r['launch_ms'], r['isolated_launch_ms'], r['frac_isolated']))"
done
