#!/bin/bash
# round 2, GPU call A: the whole -m gpu suite, the default bench line (configs[2] + companions), a kernel trace of configs[2]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r02a/pytest.log 2>&1
tail -30 gpurun_out/r02a/pytest.log
( time timeout 600 python bench.py ) > gpurun_out/r02a/bench_default.log 2>&1
tail -3 gpurun_out/r02a/bench_default.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r02a/prof_b256" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --cpu-utts 0 --companions 0 > "$GRAFT_REPO_ROOT/gpurun_out/r02a/prof_b256.log" 2>&1
tail -2 "$GRAFT_REPO_ROOT/gpurun_out/r02a/prof_b256.log"
