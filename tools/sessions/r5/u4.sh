#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for M in 'siib&estoi' 'siib&haspi&estoi'; do
for B in 16 32 64 128; do
timeout 600 python tools/pipe_time.py $B "$M" 20 plain,late,feats,early,plain 2>&1 | grep "ms/step"
done
done
