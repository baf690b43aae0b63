#!/bin/bash
# r6 call s: the x ** float32(1/6) / x ** 6 fast paths (band features, G-step glue): tests + timing of the enhancement path
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_features_gpu.py tests/test_model_gpu.py tests/test_epoch_gpu.py tests/test_varlen_gpu.py tests/test_step_parity_gpu.py tests/test_netplan_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -6
python tools/infer_time.py 128 20 plain 2>&1 | tail -1
INFLIGHT=4 python tools/infer_time.py 128 40 stream 2>&1 | tail -2
python tools/infer_time.py 128 10 stages 2>&1 | tail -1
