#!/bin/bash
# A/B inside ONE gpurun call (boxes differ by several per cent).  usage: tools/ab.sh "ENV1=a ENV2=b" "ENV1=c" ... [-- bench args]
# every quoted group is one variant; the variants are run round-robin three times.  The NELE_* switches only exist in the TEST library
# (libnele_hip_ab.so, -DNELE_AB): it is what every variant loads (NELE_LIB); the product library has one path per operation.
cd "$GRAFT_REPO_ROOT"
export NELE_LIB=${NELE_LIB:-$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip_ab.so}
variants=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do variants+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rep in 1 2 3; do for v in "${variants[@]}"; do
  env $v timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --companions 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_mfma', d['roofline']); print('%-60s' % '$v', 'ms/step', round(d['ms_per_step'],2), 'conv5 iso', round(r['isolated_launch_ms'],3), 'in-step', round(r['launch_ms'],3))"
done; done
