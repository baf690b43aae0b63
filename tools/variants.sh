#!/bin/bash
# Build variants of ONE translation unit with extra -D flags and link each into its own library (A/B on the GPU box via NELE_LIB):
#   tools/variants.sh conv16 name1:"-DFOO" name2:"-DBAR -DBAZ" ...   -> nele_gan_amd/libnele_hip.so.<name>
unit=$1; shift
cd "$(dirname "$0")/../nele_gan_amd/csrc"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable $flags -c $unit.hip -o ../../build/csrc/$unit.$name.o || exit 1
  objs=$(ls ../../build/csrc/*.o | grep -v "\.[A-Za-z0-9_]*\.o$" | grep -v "/$unit.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libnele_hip.so.$name $objs ../../build/csrc/$unit.$name.o || exit 1
  echo built libnele_hip.so.$name
done
