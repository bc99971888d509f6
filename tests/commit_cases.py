"""Helpers shared by the tests of the content-aware commit (mi_memfs_commit_layer with a ctx): tests/test_gpu_commit.py
drives it on the MI355X against the oracle, tests/hip_stub/commit_scenarios.py drives its host side on the HIP test
double.  Test infrastructure only."""
import hashlib
import io
import os
import tarfile

import numpy as np

SEED, MASK_BITS, MIN_SIZE, MAX_SIZE = 0x4D414B49, 13, 2048, 65536


def write_file(path, data, mode=0o644, mtime=None):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(data)
    os.chmod(path, mode)
    if mtime is not None:
        os.utime(path, (mtime, mtime))


def make_tree(root, seed=1, n_dirs=6, files_per_dir=9, big_every=4, mtime=1_600_000_000):
    """A tree whose files take BOTH ways into the batch: most are small (read where they are listed, a block per directory),
    every big_every-th is larger than the walk's inline limit of 16 KiB (handed to the reader threads as a path) -- so the
    order in which bytes reach the arena is not the order of the file table.  Also: empty files, a one-byte file, symlinks
    (relative and absolute), a nested directory chain, an empty directory.  Every mtime is `mtime`: a later rewrite with
    the same size can keep the second.  Returns {relpath: bytes} of the regular files."""
    rng = np.random.default_rng(seed)
    files = {}
    for d in range(n_dirs):
        sub = "d%02d" % d if d % 2 == 0 else "d%02d/nested/deeper" % d
        for k in range(files_per_dir):
            if k % big_every == big_every - 1:
                size = int(rng.integers(17_000, 300_000))
            elif k == 0:
                size = 0 if d % 3 == 0 else 1
            else:
                size = int(rng.integers(1, 16_000))
            rel = "%s/f%03d.bin" % (sub, k)
            data = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
            write_file(os.path.join(root, rel), data, 0o644 if k % 2 else 0o755, mtime)
            files[rel] = data
    os.makedirs(os.path.join(root, "empty_dir"), exist_ok=True)
    os.symlink("d00/f001.bin", os.path.join(root, "rel_link"))
    os.symlink(os.path.join(root, "d00"), os.path.join(root, "abs_link"))
    for dp, dns, fns in os.walk(root):
        os.utime(dp, (mtime, mtime))
    return files


def tar_members(raw):
    """[(name, TarInfo, bytes or None)] of a tar given as bytes, in archive order"""
    out = []
    with tarfile.open(fileobj=io.BytesIO(raw)) as tf:
        for m in tf.getmembers():
            out.append((m.name, m, tf.extractfile(m).read() if m.isfile() else None))
    return out


def oracle_chunks(O, data):
    """the chunk rows the ORACLE gives a file's bytes (one per chunk)"""
    p = O.CdcParams(SEED, MASK_BITS, MIN_SIZE, MAX_SIZE)
    a = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(0, dtype=np.uint8)
    _, rc = O.scan_batch(a, [0], [len(data)], p, True, 1, 0)
    return rc


def oracle_root(O, data):
    """the chunk root the ORACLE gives a file's bytes (Gear CDC with the default parameters, SHA-256 per chunk, root)"""
    p = O.CdcParams(SEED, MASK_BITS, MIN_SIZE, MAX_SIZE)
    a = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(0, dtype=np.uint8)
    rf, _ = O.scan_batch(a, [0], [len(data)], p, True, 1, 0)
    return rf["chunk_root"][0].tobytes()


def commit_to_bytes(fs, tmp, name, **kw):
    """fs.commit_layer(...) with the plain tar (no gzip leg) written to a file; returns (result, tar bytes)"""
    import makisu_amd as M
    path = os.path.join(str(tmp), name)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        res = fs.commit_layer(out_fd=fd, gzip_level=M.GZIP_OFF, **kw)
    finally:
        os.close(fd)
    raw = open(path, "rb").read()
    if res is not None:
        assert res["tar_digest"] == "sha256:" + hashlib.sha256(raw).hexdigest() and res["tar_bytes"] == len(raw)
    return res, raw


def proc_io():
    """(rchar, syscr) of this process: bytes and calls of read-like system calls, as the kernel counts them"""
    d = dict(ln.split(": ") for ln in open("/proc/self/io").read().splitlines())
    return int(d["rchar"]), int(d["syscr"])
