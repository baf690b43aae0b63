#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_dataio_gpu.py tests/test_epoch_gpu.py -x -q 2>&1 | tail -8
timeout 900 python tools/files_time.py 4096 256 2>&1 | grep -v "^\[W\|amdgpu.ids" | cut -c1-900
