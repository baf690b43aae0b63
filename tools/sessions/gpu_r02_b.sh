#!/bin/bash
# round 2, GPU call B: HASPI split tests + A/B of the split at configs[2] + trace
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02b
( timeout 900 python -m pytest tests/test_metrics_gpu.py tests/test_step_parity_gpu.py tests/test_train_gpu.py -m gpu -x -q ) > gpurun_out/r02b/pytest.log 2>&1
tail -5 gpurun_out/r02b/pytest.log
NELE_HASPI_SPLIT=0 timeout 300 python bench.py --steps 8 --warmup 2 --cpu-utts 0 --companions 0 2>/dev/null | cut -c1-250
NELE_HASPI_SPLIT=1 timeout 300 python bench.py --steps 8 --warmup 2 --cpu-utts 0 --companions 0 2>/dev/null | cut -c1-250
timeout 300 python bench.py --steps 8 --warmup 2 --cpu-utts 0 --companions 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['configs1'], d['nonperiodic'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r02b/prof_b256" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --cpu-utts 0 --companions 0 > "$GRAFT_REPO_ROOT/gpurun_out/r02b/prof_b256.log" 2>&1
