#!/usr/bin/env python3
"""The chunk pass of a C2-shaped batch by arena kind, load scheme and footprint: where an arena of small pieces
(mi_arena.hip) wants the quad-cooperative loads (ShaTune::coop_min_bytes_pieces).

One process per arena kind (MI_ARENA is read once); in it, for every footprint and scheme, a ctx, a batch of
`files` 64 KiB synthetic files, seven runs, the median `ms_sha_chunks` of the last five (one batch at a time).

  python tools/pieces_scheme_ab.py              # all arena kinds, as subprocesses
  MI_ARENA=pieces python tools/pieces_scheme_ab.py --one pieces32
"""
import argparse
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

KINDS = {"malloc": {"MI_ARENA": "malloc"},
         "pieces32": {"MI_ARENA": "pieces", "MI_ARENA_PIECE_MB": "32"},
         "pieces128": {"MI_ARENA": "pieces", "MI_ARENA_PIECE_MB": "128"}}
FILES = [2000, 8000, 16000, 32000, 50000, 100000]


def one(kind, files_list):
    import numpy as np
    import makisu_amd
    from makisu_amd import workloads as W
    for n in files_list:
        sh = W.c2(files_per_gpu=n)
        row = []
        for name, scheme in (("lane", makisu_amd.SHA_LOADS_LANE), ("coop", makisu_amd.SHA_LOADS_COOP),
                             ("auto", makisu_amd.SHA_LOADS_AUTO)):
            e = makisu_amd.Engine(device=0, sha_load_scheme=scheme)
            b = e.batch(sh.n_files, sh.n_bytes)
            b.add_synthetic(sh.sizes, sh.cids, seed=sh.seed)
            ms = []
            for i in range(7):
                b.run() if i == 0 else b.rerun()
                if i >= 2:
                    ms.append(e.stats()["ms_sha_chunks"])
            b.free()
            e.close()
            row.append("%s %.4f" % (name, float(np.median(ms))))
        print("arena %-9s files %6d (%5.2f GiB)  chunk pass ms: %s" % (kind, n, sh.n_bytes / 2**30, "  ".join(row)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--one", default="")
    ap.add_argument("--files", default="")
    a = ap.parse_args()
    files = [int(x) for x in a.files.split(",")] if a.files else FILES
    if a.one:
        return one(a.one, files)
    for kind, env in KINDS.items():
        e = dict(os.environ)
        e.pop("MI_ARENA", None)
        e.pop("MI_ARENA_PIECE_MB", None)
        e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", kind, "--files", ",".join(map(str, files))],
                       env=e, check=False, timeout=600)


if __name__ == "__main__":
    main()
