#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python scratch/soak_prof.py 1 2>&1 | grep -v "amdgpu.ids" | grep "ms/step"
timeout 600 python scratch/soak_prof.py 2 2>&1 | grep -v "amdgpu.ids" | grep "ms/step"
timeout 600 python scratch/soak_prof.py 8 2>&1 | grep -v "amdgpu.ids" | tail -45 | cut -c1-200
