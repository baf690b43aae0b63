#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -4
