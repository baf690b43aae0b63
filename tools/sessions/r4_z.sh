#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
timeout 600 python bench.py --cpu-utts 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', round(d['ms_per_step'],2), 'configs1', {k: round(d['configs1'][k],3) for k in ('ms_per_step','ms_per_step_plain')}, 'hp', {k: round(d['headline_pipelined'][k],3) for k in ('ms_per_step','ms_per_step_plain')})"
done
