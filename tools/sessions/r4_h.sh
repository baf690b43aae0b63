#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4h
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r4h/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r4h/smoke.txt
python bench.py --companions 0 --cpu-utts 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['same_as'], d['roofline']['ms_per_step_by_kernel'])" | tee gpurun_out/r4h/bench.txt
