#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_dataio_gpu.py tests/test_eval_gpu.py -x -q 2>&1 | tail -8
