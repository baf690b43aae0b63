#!/bin/bash
# SIIB spectra kernel variants at the non-periodic length (kernel time under rocprofv3)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "" $@; do
  export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip.so$v
  cd /tmp; rm -rf /tmp/psp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psp -- python3 $GRAFT_REPO_ROOT/tools/siib_ab.py 256 63871 > /tmp/psp.log 2>&1
  cd $GRAFT_REPO_ROOT; echo "lib=$v $(python tools/kstats.py /tmp/psp 6 30 2>&1 | grep spec_wave) $(grep scores /tmp/psp.log | cut -c1-60)"
done
