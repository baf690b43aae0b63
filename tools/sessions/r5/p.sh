#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/prof_round.sh r05 2>&1 | tail -3
bash tools/prof_infer.sh r05 2>&1 | tail -6
cd $GRAFT_REPO_ROOT
python tools/eigh_time.py 256 420 5 2>&1 | tail -1 > gpurun_out/r05_eigh_standalone.txt; cat gpurun_out/r05_eigh_standalone.txt
timeout 1500 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; tail -2 gpurun_out/r05_bench_default.err
