#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4i
export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip_ab.so
(NELE_WGRAD_DMA_TH=2 python tools/wgrad_check.py; NELE_WGRAD_DMA_TH=4 python tools/wgrad_check.py; NELE_WGRAD_DMA_TH=2 python tools/wgrad_check.py; NELE_WGRAD_DMA_TH=4 python tools/wgrad_check.py) 2>&1 | grep -E "conv5|sum" | tee gpurun_out/r4i/wgrad.txt
