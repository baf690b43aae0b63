#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_metrics_gpu.py -m gpu -x -q -s -k "resident" 2>&1 | grep -v "^$" | tail -5
