#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip_ab.so
for th in 0 84 83 0 84 83; do
NELE_CONV16_TH=$th timeout 600 python bench.py --steps 10 --warmup 3 --companions 0 --cpu-utts 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_mfma']
print('TH=$th: %.2f ms/step  conv5 fwd in-step %.3f ms, isolated %.3f ms (frac %.3f)' % (d['ms_per_step'], r['launch_ms'], r['isolated_launch_ms'], r['frac_isolated']))"
done
