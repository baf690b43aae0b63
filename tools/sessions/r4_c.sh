#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4c
(NELE_WGRAD_DMA=0 python tools/wgrad_check.py; NELE_WGRAD_DMA=1 python tools/wgrad_check.py; NELE_WGRAD_DMA_TH=2 python tools/wgrad_check.py;  python tools/wgrad_check.py 32) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c/wgrad.txt
python -m pytest tests/test_model_gpu.py tests/test_step_parity_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r4c/tests.txt
