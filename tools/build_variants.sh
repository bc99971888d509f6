#!/bin/bash
# Dev tool: builds libmakisu_mi variants of sha256.hip with different -D knobs into tools/bin/
# usage: tools/build_variants.sh name "-DX=1 -DY=2" [source-file]
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; src=${3:-makisu_amd/csrc/sha256.hip}
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -fno-gpu-rdc \
  -mllvm -amdgpu-atomic-optimizer-strategy=None -Imakisu_amd/csrc $flags -c $src -o tools/bin/sha256_$name.o
objs=$(ls makisu_amd/_obj/*.o | grep -v sha256.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -Wl,--no-undefined $objs tools/bin/sha256_$name.o -ldl -lpthread -lz -o tools/bin/libmi_$name.so
echo tools/bin/libmi_$name.so
