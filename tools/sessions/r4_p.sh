#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4p
for rep in 1 2; do for v in base s0 s3 s4 s5 s4p; do echo "== $v"; NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip.so.$v timeout 200 python tools/eigh_time.py 256 420 5 2>&1 | tail -2; done; done | tee gpurun_out/r4p/s1split.txt
