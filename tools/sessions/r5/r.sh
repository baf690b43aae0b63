#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_features_gpu.py tests/test_epoch_gpu.py tests/test_varlen_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/infer_time.py 128 20 stages
INFLIGHT=3 python tools/infer_time.py 128 60 stream
INFLIGHT=3 python tools/infer_time.py 256 60 stream
