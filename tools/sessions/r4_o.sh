#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4o
timeout 600 python -m pytest tests/test_metrics_gpu.py -x -q -m gpu -k "eigensolver or repaired" 2>&1 | tail -3
NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip.so.prof timeout 120 python tools/eigh_time.py 256 420 2 2>&1 | grep -v amdgpu | tee gpurun_out/r4o/prof.txt
export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip_ab.so
(NELE_EIGH_SYM=1 timeout 120 python tools/eigh_time.py 256 420 3; NELE_EIGH_SYM=0 timeout 120 python tools/eigh_time.py 256 420 3) 2>&1 | grep -v amdgpu | tee gpurun_out/r4o/eigh.txt
