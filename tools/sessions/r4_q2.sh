#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/eigh_time.py 256 420 5 2>&1 | tail -1
python -m pytest tests/test_metrics_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/ab.sh "NELE_X=1" 2>&1 | tail -3
