#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4f
bash tools/ab.sh "NELE_EIGH_P4_BATCH=32" "NELE_EIGH_P4_BATCH=64" "NELE_EIGH_P4_BATCH=16" 2>&1 | tee gpurun_out/r4f/ab_batch.txt
