#!/bin/bash
# The CPU suite against a build of the library with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST code
# (device code is compiled as usual: -fno-gpu-sanitize).  No GPU needed: everything `-m "not gpu"` reaches -- walks, the
# layer merge and diff, copy ops, the tar reader and writer, the layer pipeline, the codecs -- runs instrumented.
#   tools/asan_host_tests.sh [pytest args]        (MI_ASAN_TARGET="<files>": those test files instead of all of tests/)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${MI_ASAN_DIR:-/tmp/mi_asan}
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fno-gpu-rdc -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -fno-omit-frame-pointer"
for f in mi_api gear_cdc sha256 tables crc32 mi_tree mi_comm mi_index mi_alloc mi_arena mi_tar mi_stage mi_layer mi_memfs; do
    extra=""
    [ "$f" = sha256 ] && extra="-mllvm -amdgpu-atomic-optimizer-strategy=None"
    /opt/rocm/bin/hipcc $FLAGS $extra -c "$ROOT/makisu_amd/csrc/$f.hip" -o "$OUT/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -fsanitize=address,undefined -fno-gpu-sanitize \
    -shared-libsan "$OUT"/*.o -ldl -lpthread -lz -o "$OUT/libmakisu_mi.so"
cd "$ROOT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    MAKISU_MI_LIB="$OUT/libmakisu_mi.so" python -m pytest ${MI_ASAN_TARGET:-tests} -q -m "not gpu" -p no:cacheprovider "$@"
