#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dataio_gpu.py tests/test_epoch_gpu.py tests/test_features_gpu.py -x -q -m gpu 2>&1 | tail -4
python tools/files_sweep.py 2>&1 | grep -v amdgpu.ids
