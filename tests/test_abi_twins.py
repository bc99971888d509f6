"""The product ABI and its oracle twins through ONE harness ("CPU-oracle twins with
identical signatures (mi_ref_*) so the parity harness calls both").

`Lib` binds either library by prefix: the ctypes signatures are written once.  CPU: the twins load,
export the batch path and answer correctly on their own.  GPU: the same call sequence against
libmakisu_mi.so and libmi_oracle.so, every output struct compared byte for byte.
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import makisu_amd
from makisu_amd import Config, FILE_DTYPE, CHUNK_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWINS = ["abi_version", "config_default", "ctx_create", "ctx_destroy", "last_error", "batch_begin", "batch_add_bytes",
         "batch_add_path", "batch_run", "batch_counts", "batch_files", "batch_chunks", "batch_free", "dedup_mark",
         "sha256_many"]


class Lib:
    def __init__(self, cdll, prefix):
        vp, u64, u64p = C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)
        sigs = {"abi_version": ([], C.c_int), "config_default": ([C.POINTER(Config)], C.c_int),
                "ctx_create": ([C.POINTER(Config), C.POINTER(vp)], C.c_int), "ctx_destroy": ([vp], C.c_int),
                "last_error": ([vp], C.c_char_p), "batch_begin": ([vp, u64, u64, C.POINTER(vp)], C.c_int),
                "batch_add_bytes": ([vp, vp, u64, u64], C.c_int), "batch_add_path": ([vp, C.c_char_p, u64, u64], C.c_int),
                "batch_run": ([vp], C.c_int), "batch_counts": ([vp, u64p, u64p, u64p], C.c_int),
                "batch_files": ([vp, vp, u64], C.c_int), "batch_chunks": ([vp, vp, u64], C.c_int),
                "batch_free": ([vp], C.c_int), "dedup_mark": ([vp, vp, u64, vp, u64p], C.c_int),
                "sha256_many": ([vp, vp, u64p, u64p, u64, vp], C.c_int)}
        for name, (args, res) in sigs.items():
            fn = getattr(cdll, prefix + name)             # AttributeError = a twin is missing
            fn.argtypes, fn.restype = args, res
            setattr(self, name, fn)


def _scan(lib, blobs, paths, **cfg_over):
    cfg = Config()
    assert lib.config_default(C.byref(cfg)) == 0
    for k, v in cfg_over.items():
        setattr(cfg, k, v)
    ctx, b = C.c_void_p(), C.c_void_p()
    assert lib.ctx_create(C.byref(cfg), C.byref(ctx)) == 0
    assert lib.batch_begin(ctx, 0, 0, C.byref(b)) == 0
    for i, x in enumerate(blobs):
        buf = (C.c_uint8 * max(len(x), 1)).from_buffer_copy(x or b"\0")
        assert lib.batch_add_bytes(b, buf if x else None, len(x), 100 + i) == 0
    for i, (p, n) in enumerate(paths):
        assert lib.batch_add_path(b, os.fsencode(p), n, 200 + i) == 0
    assert lib.batch_add_path(b, b"/nonexistent/file", 1, 0) == -5          # MI_ERR_IO on both sides
    assert lib.batch_run(b) == 0
    nf, nc, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert lib.batch_counts(b, C.byref(nf), C.byref(nc), C.byref(nb)) == 0
    files = np.zeros(max(nf.value, 1), dtype=FILE_DTYPE)
    chunks = np.zeros(max(nc.value, 1), dtype=CHUNK_DTYPE)
    assert lib.batch_files(b, files.ctypes.data, nf.value - 1) == -7 if nf.value else True   # MI_ERR_CAPACITY
    assert lib.batch_files(b, files.ctypes.data, nf.value) == 0
    assert lib.batch_chunks(b, chunks.ctypes.data, nc.value) == 0
    assert lib.batch_free(b) == 0
    assert lib.ctx_destroy(ctx) == 0
    return (nf.value, nc.value, nb.value), files[:nf.value].copy(), chunks[:nc.value].copy()


def _inputs(tmp_path):
    rng = np.random.default_rng(12)
    blobs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (70000, 0, 1, 300000, 4097)]
    blobs.append(blobs[0])                                                    # a duplicate file
    paths = []
    for i, n in enumerate((65536, 10)):
        p = tmp_path / ("in%d" % i)
        p.write_bytes(rng.integers(0, 256, n + 5, dtype=np.uint8).tobytes())  # longer than `size`: CopyN cuts
        paths.append((str(p), n))
    return blobs, paths


def test_oracle_twins_exist_and_work(oracle, tmp_path):
    ref = Lib(oracle.lib(), "mi_ref_")
    assert ref.abi_version() == makisu_amd.load_library().mi_abi_version()
    blobs, paths = _inputs(tmp_path)
    counts, files, chunks = _scan(ref, blobs, paths, flags=makisu_amd.FLAG_FILE_SHA256)
    assert counts[0] == len(blobs) + len(paths)
    whole = blobs + [open(p, "rb").read()[:n] for p, n in paths]
    assert [bytes(r) for r in files["file_sha256"]] == [hashlib.sha256(x).digest() for x in whole]
    assert list(files["user_tag"]) == [100 + i for i in range(len(blobs))] + [200, 201]
    assert (chunks["dup_of"][chunks["file_index"] == 5] >= 0).all()           # the duplicate file
    for f in range(counts[0]):                                                # chunks tile every file
        mine = chunks[chunks["file_index"] == f]
        assert int(mine["length"].sum()) == len(whole[f])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [{}, {"flags": 1}, {"mask_bits": 10, "min_size": 256, "max_size": 8192}])
def test_product_and_twin_agree_through_one_harness(oracle, tmp_path, cfg):
    try:
        import torch  # noqa: F401  (load order: see test_gpu_parity.py)
    except ImportError:
        pass
    blobs, paths = _inputs(tmp_path)
    got = _scan(Lib(makisu_amd.load_library(), "mi_"), blobs, paths, **cfg)
    want = _scan(Lib(oracle.lib(), "mi_ref_"), blobs, paths, **cfg)
    assert got[0] == want[0]
    assert got[1].tobytes() == want[1].tobytes()                              # mi_file_result rows, every byte
    assert got[2].tobytes() == want[2].tobytes()                              # mi_chunk_result rows, every byte
