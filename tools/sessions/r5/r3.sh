#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python scratch/files_break.py 4096 2>&1 | grep -v "^\[W\|amdgpu.ids" | cut -c1-300
nproc; free -g | head -2
