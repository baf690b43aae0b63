#!/bin/bash
# A/B inside ONE gpurun call (boxes differ by several per cent).  usage: tools/ab.sh "ENV1=a ENV2=b" "ENV1=c" ... [-- bench args]
# every quoted group is one variant; the variants are run round-robin three times.
cd "$GRAFT_REPO_ROOT"
variants=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do variants+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rep in 1 2 3; do for v in "${variants[@]}"; do
  env $v timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --companions 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_mfma', d['roofline']); print('%-60s' % '$v', 'ms/step', round(d['ms_per_step'],2), 'conv5 iso', round(r['isolated_launch_ms'],3), 'in-step', round(r['launch_ms'],3))"
done; done
