#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_netplan_gpu.py -q -m gpu > gpurun_out/r6d_test.log 2>&1
tail -60 gpurun_out/r6d_test.log
timeout 600 python -m pytest tests/test_clean_cache_gpu.py tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r6d_test2.log 2>&1
tail -5 gpurun_out/r6d_test2.log
