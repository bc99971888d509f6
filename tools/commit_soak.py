"""A soak of the content-aware commit on the GPU: tests/test_gpu_commit.py's replay property (a seeded sequence of steps -- new files,
deletions, same-size same-second rewrites, a symlink retargeted -- committed with a ctx, the layers stacked again: every byte and every
root must be right after every step) over many seeds, alternately pipelined / phase by phase is a process-wide setting, so: one process
per mode.  usage: commit_soak.py [first seed = 100] [seeds = 40]      (MI_COMMIT_PIPELINE=0 for the other mode; MI_SOAK_TRUST=1: the
handle with MI_MEMFS_TRUST_CTIME; MI_COMMIT_FORCE_WINDOWS=1 MI_COMMIT_WINDOW_MB=1: as if the tree did not fit the device;
MI_SOAK_N_CTXS=k: every commit over k ctxs on the one device -- mi_memfs_commit_layer_n)"""
import os
import sys
import tempfile
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathlib  # noqa: E402

import makisu_amd as M  # noqa: E402
from oracle import mi_oracle as O  # noqa: E402
import test_gpu_commit as T  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    O.build()
    ok = 0
    k = int(os.environ.get("MI_SOAK_N_CTXS", "1"))
    more = [M.Engine(device=0, n_streams=2) for _ in range(k - 1)]
    with M.Engine(device=0) as eng0:
        eng = [eng0] + more if more else eng0
        for seed in range(first, first + n):
            tmp = pathlib.Path(tempfile.mkdtemp(prefix="mi_commit_soak_"))
            try:
                T.test_the_layers_of_a_build_replay_to_the_tree_bytes_included.__wrapped__(O, eng, tmp, seed) \
                    if hasattr(T.test_the_layers_of_a_build_replay_to_the_tree_bytes_included, "__wrapped__") \
                    else T.test_the_layers_of_a_build_replay_to_the_tree_bytes_included(O, eng, tmp, seed)
                ok += 1
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
    for e in more:
        e.close()
    print("commit soak: %d of %d seeds replayed to the tree, bytes and roots (MI_COMMIT_PIPELINE=%s, trust_ctime=%s, forced windows=%s, ctxs=%d)" %
          (ok, n, os.environ.get("MI_COMMIT_PIPELINE", "1"), os.environ.get("MI_SOAK_TRUST", "0"), os.environ.get("MI_COMMIT_FORCE_WINDOWS", "0"), k))


if __name__ == "__main__":
    main()
