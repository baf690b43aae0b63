#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for k in 1 2 3 4; do
for w in 0 1; do
NELE_WGRAD_QUEUES=$w timeout 600 python bench.py --steps 10 --warmup 3 --companions 0 --cpu-utts 0 --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_queues=$w B=256: %.2f ms/step' % d['ms_per_step'])"
done
done
for w in 0 1; do
NELE_WGRAD_QUEUES=$w timeout 600 python bench.py --steps 10 --warmup 3 --companions 0 --cpu-utts 0 --no-isolated --batch 128 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_queues=$w B=128: %.2f ms/step' % d['ms_per_step'])"
NELE_WGRAD_QUEUES=$w timeout 600 python tools/pipe_time.py 32 'siib&estoi' 20 plain 2>&1 | grep ms/step
done
