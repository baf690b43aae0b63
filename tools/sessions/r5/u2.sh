#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python tools/soak.py 400 64 2>&1 | grep -v "amdgpu.ids" | tail -12
df -h /dev/shm | tail -1
