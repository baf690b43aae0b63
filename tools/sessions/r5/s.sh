#!/bin/bash
cd $GRAFT_REPO_ROOT
python scratch/pow_check.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_features_gpu.py tests/test_epoch_gpu.py tests/test_varlen_gpu.py tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/infer_time.py 128 20 stages
INFLIGHT=3 python tools/infer_time.py 128 60 stream
