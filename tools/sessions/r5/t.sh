#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_metrics_gpu.py tests/test_step_parity_gpu.py tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab.sh "NELE_X=0" 2>&1 | tail -3
