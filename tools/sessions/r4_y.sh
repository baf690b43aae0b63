#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "" .bar0; do
  echo "lib=$v $(NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip.so$v timeout 100 python tools/eigh_time.py 256 420 5 2>&1 | tail -1)"
done; done
timeout 600 python -m pytest tests/test_metrics_gpu.py -x -q -m gpu -k "eig or sym or exchange or repair" 2>&1 | tail -2
