#!/bin/bash
# per-kernel register / spill / occupancy table of one .hip file: tools/kres.sh nele_gan_amd/csrc/conv16.hip [filter]
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Inele_gan_amd/csrc -Rpass-analysis=kernel-resource-usage -c $f $KRES_FLAGS -o /tmp/kres.o 2>&1 | \
 awk '/Function Name/ {n=$5} / VGPRs:/ {v=$4} /AGPRs:/ {a=$4} /VGPRs Spill/ {sp=$5} /SGPRs:/ {sg=$4} /Occupancy/ {o=$5} /LDS Size/ {print n, "vgpr", v, "agpr", a, "sgpr", sg, "spill", sp, "occ", o}' | c++filt | grep -E "$pat"
