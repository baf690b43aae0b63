#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_step_parity_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/g_time.py 256 251 bf16
python tools/g_time.py 128 501 bf16
python tools/g_time.py 32 251 bf16
