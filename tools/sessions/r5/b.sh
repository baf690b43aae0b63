#!/bin/bash
# round 5, call b: streamed inference (tests, sweep of batch size x batches in flight)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_epoch_gpu.py tests/test_varlen_gpu.py tests/test_features_gpu.py -m gpu -x -q 2>&1 | tail -5
for B in 64 128 256; do
  python tools/infer_time.py $B 30 plain
  for n in 2 3 4; do INFLIGHT=$n python tools/infer_time.py $B 60 stream; done
done
