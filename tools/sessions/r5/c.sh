#!/bin/bash
# round 5, call c: fused generator layer kernel: numerics, then inference timings
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "fused or bf16_operand or golden" 2>&1 | tail -15
python tools/infer_time.py 128 20 stages
python tools/infer_time.py 128 30 plain
INFLIGHT=3 python tools/infer_time.py 128 60 stream
INFLIGHT=3 python tools/infer_time.py 256 60 stream
