#!/bin/bash
# Dev tool: builds a libmakisu_mi variant with ONE source compiled with extra -D knobs into tools/bin/
# usage: tools/build_variants.sh name "-DX=1 -DY=2" [source-file (default makisu_amd/csrc/sha256.hip)]
# (run `python -m makisu_amd.build` first: the other objects come from makisu_amd/_obj)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; src=${3:-makisu_amd/csrc/sha256.hip}
base=$(basename $src .hip)
mkdir -p tools/bin
extra=""
[ "$base" = sha256 ] && extra="-mllvm -amdgpu-atomic-optimizer-strategy=None"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -fno-gpu-rdc \
  $extra -Imakisu_amd/csrc $flags -c $src -o tools/bin/${base}_$name.o
objs=$(ls makisu_amd/_obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -Wl,--no-undefined $objs tools/bin/${base}_$name.o -ldl -lpthread -lz -o tools/bin/libmi_$name.so
echo tools/bin/libmi_$name.so
