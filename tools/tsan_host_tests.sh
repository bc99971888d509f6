#!/bin/bash
# The threaded host code (parallel walk, the layer writer's tee / sink threads / deflate pool, the tar reader, and -- on
# the HIP test double of tests/hip_stub -- the reader threads, pinned slabs and stream ordering of the host-fed path) under
# ThreadSanitizer -- the reference runs its tests with `go test -race`.  No GPU needed.  Reports go to $OUT/report.*;
# the script fails if there is one.  (MI_WALK_UNSHARE=0: the walk's directory readers keep the process's descriptor table --
# the detector models descriptors per process and reports two threads' private "fd 4" as one.)
#   tools/tsan_host_tests.sh [pytest args; default: the host test files]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${MI_TSAN_DIR:-/tmp/mi_tsan}
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
mkdir -p "$OUT"
rm -f "$OUT"/report.*
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fno-gpu-rdc -fsanitize=thread -fno-gpu-sanitize -shared-libsan -fno-omit-frame-pointer"
for f in mi_api gear_cdc sha256 tables crc32 mi_tree mi_comm mi_index mi_alloc mi_arena mi_tar mi_stage mi_layer mi_memfs; do
    extra=""
    [ "$f" = sha256 ] && extra="-mllvm -amdgpu-atomic-optimizer-strategy=None"
    /opt/rocm/bin/hipcc $FLAGS $extra -c "$ROOT/makisu_amd/csrc/$f.hip" -o "$OUT/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -fsanitize=thread -fno-gpu-sanitize -shared-libsan \
    "$OUT"/*.o -ldl -lpthread -lz -o "$OUT/libmakisu_mi.so"
cd "$ROOT"
ARGS=("$@")
[ ${#ARGS[@]} -eq 0 ] && ARGS=(tests/test_host_walk.py tests/test_host_layer.py tests/test_host_layer_properties.py
                                tests/test_host_tar.py tests/test_host_tar_properties.py tests/test_host_copy_ops.py
                                tests/test_host_apply_properties.py tests/test_host_diff_properties.py
                                tests/test_host_hip_double.py tests/test_host_commit_double.py)
MI_WALK_UNSHARE=0 LD_PRELOAD=$RT TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:exitcode=0:log_path="$OUT/report" \
    MAKISU_MI_LIB="$OUT/libmakisu_mi.so" python -m pytest -q -p no:cacheprovider "${ARGS[@]}"
if ls "$OUT"/report.* >/dev/null 2>&1; then
    grep -h "SUMMARY" "$OUT"/report.* | sort | uniq -c
    exit 1
fi
echo "ThreadSanitizer: no reports"
