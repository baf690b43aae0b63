#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_varlen_gpu.py tests/test_step_parity_gpu.py -x -q 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/t4prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/kstats.py gpurun_out/t4prof 4 120 | grep -E "gap|conv16_kernel<4|total" | cut -c1-150
for k in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --companions 0 --cpu-utts 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_mfma']
print('%.2f ms/step  conv5 fwd in-step %.3f ms, isolated %.3f ms (frac %.3f)' % (d['ms_per_step'], r['launch_ms'], r['isolated_launch_ms'], r['frac_isolated']))"
done
