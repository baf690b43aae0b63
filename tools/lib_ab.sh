#!/bin/bash
# A/B of library builds (tools/variants.sh) on one stand-alone script under rocprofv3: tools/lib_ab.sh "<kernel pattern>" "<suffixes, '' = the default build>" script args...
pat=$1; shift; sufs=$1; shift
for v in $sufs; do
  [ "$v" = "-" ] && v=""
  export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip.so$v
  echo "== lib=$v"
  bash tools/prof_one.sh "$@" 2>&1 | grep -i "scores\|per call\|$pat"
done
