"""A soak of SPLIT FILES in the commit over several ctxs (files of 256 MiB and more go over the GPUs as parts; here the
threshold is lowered to 2 MiB so that a seed's tree holds several of them and a box's one GPU carries all the ctxs).

Per seed: a tree of 3-6 files of 2-24 MiB (sizes not aligned to anything) and a few small ones; contents by class --
random bytes; ALL ZEROS and a 4 KiB period (no content cut ever re-synchronises a part's halo: every boundary is settled by
the rounds of the parts protocol, forced cuts at max_size all the way); the same two behind a random prefix of odd length ("shifted": the forced cuts lie on a grid that
begins where the prefix's last content cut fell, so every part's assumed entry is wrong and its true one is known only when
its predecessor's cuts are final -- as many rounds as the file has parts); random with a long stretch repeated at another
offset -- committed over 2, 3, 4 or 8 ctxs.  Held against: the header-only commit's tar (TarDigest and size), the ORACLE's
root of every whole file, the byte check's counts (every layer file verified); then one byte of a split file is flipped at
a random offset, size and second kept: the next commit's layer is that file alone, its root the oracle's again.

usage: split_soak.py [first seed = 1] [seeds = 24]      (MI_COMMIT_PIPELINE=0 for the phase-by-phase commit)"""
import os
import pathlib
import shutil
import sys
import tempfile

os.environ.setdefault("MI_COMMIT_SPLIT_MIB", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import makisu_amd as M  # noqa: E402
from commit_cases import oracle_root, write_file  # noqa: E402
from oracle import mi_oracle as O  # noqa: E402

MTIME = 1_600_000_000
MIB = 1 << 20


def content(rng, size, kind):
    if kind == "zeros":
        return bytes(size)
    if kind == "period":
        return (rng.integers(0, 256, 4096, dtype=np.uint8).tobytes() * (size // 4096 + 1))[:size]
    if kind in ("shifted zeros", "shifted period"):
        pre = rng.integers(0, 256, int(rng.integers(1, 70_000)) | 1, dtype=np.uint8).tobytes()
        return (pre + content(rng, size, kind[len("shifted "):]))[:size]
    a = rng.integers(0, 256, size, dtype=np.uint8)
    if kind == "repeat" and size > 6 * MIB:                    # a stretch that occurs twice, the second time across a part boundary
        n = int(rng.integers(MIB, 2 * MIB))
        src = int(rng.integers(0, size // 2 - n))
        dst = int(rng.integers(size // 2, size - n))
        a[dst:dst + n] = a[src:src + n]
    return a.tobytes()


def one_seed(seed, engines, tmp):
    rng = np.random.default_rng(seed)
    root = str(tmp / "root")
    files = {}
    kinds = ["random", "zeros", "period", "repeat", "shifted zeros", "shifted period", "shifted zeros"]
    for i in range(int(rng.integers(3, 7))):
        size = int(rng.integers(2 * MIB, 24 * MIB)) + int(rng.integers(0, 4096))
        files["big/f%d.bin" % i] = content(rng, size, kinds[int(rng.integers(0, len(kinds)))])
    for i in range(int(rng.integers(2, 9))):
        files["etc/s%d" % i] = rng.integers(0, 256, int(rng.integers(0, 200_000)), dtype=np.uint8).tobytes()
    for rel, data in files.items():
        write_file(os.path.join(root, rel), data, 0o644, MTIME)
    for dp, _, _ in os.walk(root):
        os.utime(dp, (MTIME, MTIME))
    with M.MemFS(root) as fs, M.MemFS(root) as plain:
        res = fs.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF, engine=engines)
        res0 = plain.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF)
        st = res["stats"]
        assert res["tar_digest"] == res0["tar_digest"] and res["tar_bytes"] == res0["tar_bytes"], (seed, "tar")
        n_big = sum(1 for r in files if r.startswith("big/"))
        assert st["n_split_files"] == n_big and st["n_ctxs"] == len(engines), (seed, st)
        assert st["n_verified_files"] == len(files) and st["n_refetched"] == 0, (seed, st)
        by = {e["relpath"]: e for e in res["layer"]}
        for rel, data in files.items():
            assert by[rel]["root"] == oracle_root(O, data), (seed, rel, len(data))
        rel = "big/f%d.bin" % int(rng.integers(0, n_big))
        changed = bytearray(files[rel])
        at = int(rng.integers(0, len(changed)))
        changed[at] ^= 1 << int(rng.integers(0, 8))
        write_file(os.path.join(root, rel), bytes(changed), 0o644, MTIME)
        os.utime(os.path.join(root, "big"), (MTIME, MTIME))
        res = fs.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF, engine=engines)
        assert [e["relpath"] for e in res["layer"] if e["kind"] == M.KIND_FILE] == [rel], (seed, rel, at)
        assert {e["relpath"]: e for e in res["layer"]}[rel]["root"] == oracle_root(O, bytes(changed)), (seed, rel, at)
    return n_big


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    O.build()
    pool = [M.Engine(device=0, n_streams=2) for _ in range(8)]
    ok = split = 0
    try:
        for seed in range(first, first + n):
            k = (2, 3, 4, 8)[seed % 4]
            tmp = pathlib.Path(tempfile.mkdtemp(prefix="mi_split_soak_"))
            try:
                split += one_seed(seed, pool[:k], tmp)
                ok += 1
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
    finally:
        for e in pool:
            e.close()
    print("split soak: %d of %d seeds (%d split files over 2 / 3 / 4 / 8 ctxs): the header-only tar, the oracle's roots of the whole files, "
          "a flipped bit caught (MI_COMMIT_PIPELINE=%s, MI_COMMIT_SPLIT_MIB=%s)"
          % (ok, n, split, os.environ.get("MI_COMMIT_PIPELINE", "1"), os.environ["MI_COMMIT_SPLIT_MIB"]))


if __name__ == "__main__":
    main()
