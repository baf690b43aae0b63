#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_varlen_gpu.py tests/test_dataio_gpu.py tests/test_epoch_gpu.py -m gpu -x -q 2>&1 | tail -3
python scratch/infer_files.py 2>&1 | grep -v amdgpu | tail -3
