#!/bin/bash
cd $GRAFT_REPO_ROOT
CH_B=32 python tools/chains.py 2>&1 | grep -v amdgpu | tail -12
CH_B=128 CH_METRICS='siib&haspi&estoi' python tools/chains.py 2>&1 | grep -v amdgpu | tail -16
