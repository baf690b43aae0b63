#!/bin/bash
# A/B: matrices per cluster launch inside the step (32 -> 64 per launch = half chip, 64 -> 128 = whole chip), alternating
cd "$GRAFT_REPO_ROOT"
export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip_ab.so
for rep in 1 2 3; do
for pb in 32 64; do
  for L in 64000 63871; do
    NELE_EIGH_P4_BATCH=$pb timeout 600 python bench.py --steps 10 --warmup 3 --companions 0 --cpu-utts 0 --no-isolated --length $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_f64',{})
print('p4_batch $pb L $L: %.2f ms/step  eigh launch %.3f ms x %s, frac %.4f, repaired %s' % (d['ms_per_step'], r.get('launch_ms',0), r.get('launches_per_step'), r.get('frac',0), d['step_status']['eigh_repaired']))"
  done
done
done
