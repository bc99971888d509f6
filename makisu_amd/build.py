"""Builds makisu_amd/libmakisu_mi.so (HIP kernels + C ABI) for gfx950 with hipcc.

In-tree on purpose: the built .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmakisu_mi.so")
SOURCES = ["mi_api.hip", "gear_cdc.hip", "sha256.hip", "tables.hip"]
HEADERS = ["mi_common.h", os.path.join("..", "..", "include", "makisu_mi.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-result", "-fgpu-rdc" if False else "-fno-gpu-rdc"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
