#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "adam" 2>&1 | grep -v "^$" | tail -40
