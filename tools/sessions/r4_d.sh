#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4d
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r4d/tests.txt
python tools/host_time.py 32 2>&1 | grep -v amdgpu | head -6 | tee gpurun_out/r4d/host32.txt
python tools/eigh_time.py 256 420 3 2>&1 | grep -v amdgpu | tee gpurun_out/r4d/eigh.txt
python bench.py --companions 0 --cpu-utts 0 2>/dev/null | tee gpurun_out/r4d/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['same_as'], d['roofline']['ms_per_step_by_kernel'], d['roofline']['frac'])"
