"""Builds makisu_amd/libmakisu_mi.so (HIP kernels + C ABI) for gfx950 with hipcc.

In-tree on purpose: the built .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmakisu_mi.so")
SOURCES = ["mi_api.hip", "gear_cdc.hip", "sha256.hip", "tables.hip", "crc32.hip", "mi_tree.hip", "mi_comm.hip",
           "mi_index.hip", "mi_alloc.hip", "mi_arena.hip", "mi_tar.hip", "mi_stage.hip", "mi_layer.hip", "mi_memfs.hip"]
HEADERS = ["mi_common.h", "mi_internal.h", "mi_local.h", "host_sha256.h", "mi_hostpath.h", "mi_memtree.h", os.path.join("..", "..", "include", "makisu_mi.h"),
           os.path.join("..", "..", "include", "makisu_mi_host.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result",
         "-fno-gpu-rdc"]
# sha256.hip aggregates its dequeue atomic per wave by hand and consumes the result one
# iteration later; LLVM's atomic optimizer would wrap it in its own reduction + an immediate
# wait for the result, which stalls the wave for a memory round trip per dequeue.
EXTRA = {"sha256.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}
OBJ_DIR = os.path.join(HERE, "_obj")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-Wl,--no-undefined"] + objs + ["-ldl", "-lpthread", "-lz", "-o", LIB]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
