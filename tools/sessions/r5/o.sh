#!/bin/bash
cd $GRAFT_REPO_ROOT
export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip_ab.so
for B in 8 16 32 64; do for S in 1 0; do echo "B=$B SYM=$S: $(NELE_EIGH_SYM=$S python tools/eigh_time.py $B 420 5 2>&1 | tail -1)"; done; done
for S in 1 0; do echo "chains B=32 SYM=$S"; NELE_EIGH_SYM=$S CH_B=32 python tools/chains.py 2>&1 | grep -E "ms/step|B_done|end "; done
