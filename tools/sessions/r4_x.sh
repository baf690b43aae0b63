#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_epoch_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated --epoch-equivalent 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', round(d['ms_per_step'],2), 'epoch_equivalent', d['epoch_equivalent'])"
