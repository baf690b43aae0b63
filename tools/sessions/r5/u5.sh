#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for B in 32 64 128; do
PIPE_LEN=63871 timeout 600 python tools/pipe_time.py $B 'siib&haspi&estoi' 20 plain,late,early,plain 2>&1 | grep "ms/step"
done
PIPE_LEN=63871 timeout 600 python tools/pipe_time.py 64 'siib&estoi' 20 plain,early 2>&1 | grep "ms/step"
PIPE_LEN=63871 timeout 600 python tools/pipe_time.py 64 'haspi' 20 plain,early 2>&1 | grep "ms/step"
