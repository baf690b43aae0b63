#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4e
export PYTHONFAULTHANDLER=1
timeout 200 python -u tools/graph_probe.py 32 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/r4e/probe32.txt
timeout 200 python -u tools/graph_probe.py 128 'siib&haspi&estoi' 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/r4e/probe128.txt
timeout 200 python -u tools/graph_probe.py 256 'siib&haspi&estoi' 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/r4e/probe256.txt
