#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 100 python tools/eigh_time.py 256 420 5 2>&1 | tail -1
timeout 100 python tools/eigh_time.py 32 420 5 2>&1 | tail -1
NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip_ab.so NELE_EIGH_XCH_KEEP=0 timeout 100 python tools/eigh_time.py 256 420 5 2>&1 | tail -1
NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip_ab.so NELE_EIGH_XCH_KEEP=1 timeout 100 python tools/eigh_time.py 256 420 5 2>&1 | tail -1
timeout 900 python -m pytest tests/test_metrics_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 200 python tools/pipe_time.py 32 2>&1 | grep -E "plain|early" | tail -2
