#!/bin/bash
# A/B of the Sturm-count kernel's lanes per eigenvalue (tools/variants.sh eigh nl2:"-DEG_NL=2" ...): kernel time under rocprofv3 at B = 256 and B = 32
for B in 256 32; do
for v in "" .nl4 .nl2 .nl1; do
  export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip.so$v
  echo "== B=$B lib=$v"
  bash tools/prof_one.sh tools/siib_ab.py $B 2>&1 | grep -i "scores\|bisect\|per call"
done; done
