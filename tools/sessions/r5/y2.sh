#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for k in 1 2 3; do
timeout 600 python bench.py --steps 10 --warmup 3 --companions 0 --cpu-utts 0 --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=256: %.2f ms/step' % d['ms_per_step'])"
done
timeout 600 python bench.py --steps 10 --warmup 3 --companions 0 --cpu-utts 0 --no-isolated --length 63871 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=256 L=63871: %.2f ms/step' % d['ms_per_step'])"
timeout 600 python bench.py --steps 10 --warmup 3 --companions 0 --cpu-utts 0 --no-isolated --batch 128 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=128: %.2f ms/step' % d['ms_per_step'])"
timeout 900 python -m pytest tests/test_step_parity_gpu.py tests/test_epoch_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -2
