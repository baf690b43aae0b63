#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4f
bash tools/ab.sh "NELE_EIGH_SYM=0" "NELE_EIGH_SYM=1" 2>&1 | tee gpurun_out/r4f/ab_sym.txt
