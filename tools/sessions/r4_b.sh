#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4b
(NELE_WGRAD_DMA=0 python tools/wgrad_check.py; NELE_WGRAD_DMA=1 python tools/wgrad_check.py; python tools/wgrad_check.py 32; NELE_WGRAD_DMA=0 python tools/wgrad_check.py 32) > gpurun_out/r4b/wgrad.txt 2>&1
cat gpurun_out/r4b/wgrad.txt | grep -v amdgpu.ids
python -m pytest tests/test_model_gpu.py tests/test_step_parity_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r4b/tests.txt
bash tools/ab.sh "NELE_WGRAD_DMA=0" "NELE_WGRAD_DMA=1" 2>&1 | tee gpurun_out/r4b/ab.txt
