"""ctypes loader for the CPU oracle (oracle/libmi_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from makisu_amd/ (the product path).
See oracle/mi_oracle.h for what each function restates and which reference
lines it follows.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI_ORACLE_LIB: another build of the oracle (tools/mutate_host.py --oracle holds tests/test_oracle.py against mutants of it)
_LIB_PATH = os.environ.get("MI_ORACLE_LIB") or os.path.join(_HERE, "libmi_oracle.so")


def build(force=False):
    if os.environ.get("MI_ORACLE_LIB"):
        return _LIB_PATH
    deps = [os.path.join(_HERE, f) for f in ("mi_oracle.c", "mi_oracle_abi.c", "mi_oracle.h", "Makefile")]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "makisu_mi.h"))
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(d) for d in deps)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmi_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class CdcParams(C.Structure):
    _fields_ = [("gear_seed", C.c_uint64), ("mask_bits", C.c_uint32),
                ("min_size", C.c_uint32), ("max_size", C.c_uint32)]


class RefChunk(C.Structure):
    _fields_ = [("file_index", C.c_uint64), ("offset", C.c_uint64), ("length", C.c_uint32),
                ("dup_of", C.c_int64), ("sha256", C.c_uint8 * 32)]


class RefFile(C.Structure):
    _fields_ = [("n_chunks", C.c_uint64), ("first_chunk", C.c_uint64),
                ("chunk_root", C.c_uint8 * 32), ("file_sha256", C.c_uint8 * 32),
                ("crc32", C.c_uint32)]


CHUNK_DTYPE = np.dtype({"names": ["file_index", "offset", "length", "dup_of", "sha256"],
                        "formats": ["<u8", "<u8", "<u4", "<i8", ("u1", 32)],
                        "offsets": [RefChunk.file_index.offset, RefChunk.offset.offset,
                                    RefChunk.length.offset, RefChunk.dup_of.offset,
                                    RefChunk.sha256.offset],
                        "itemsize": C.sizeof(RefChunk)})
FILE_DTYPE = np.dtype({"names": ["n_chunks", "first_chunk", "chunk_root", "file_sha256", "crc32"],
                       "formats": ["<u8", "<u8", ("u1", 32), ("u1", 32), "<u4"],
                       "offsets": [RefFile.n_chunks.offset, RefFile.first_chunk.offset,
                                   RefFile.chunk_root.offset, RefFile.file_sha256.offset,
                                   RefFile.crc32.offset],
                       "itemsize": C.sizeof(RefFile)})

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        u64p = C.POINTER(C.c_uint64)
        L.mi_ref_sha256.argtypes = [C.c_void_p, C.c_size_t, u8p, C.c_int]
        L.mi_ref_sha256.restype = None
        L.mi_ref_have_shani.restype = C.c_int
        L.mi_ref_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.mi_ref_crc32.restype = C.c_uint32
        L.mi_ref_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.mi_ref_crc32_combine.restype = C.c_uint32
        L.mi_ref_gear_table.argtypes = [C.c_uint64, u64p]
        L.mi_ref_gear_table.restype = None
        L.mi_ref_gear_candidates.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                             u64p, C.c_uint32, u64p, C.c_size_t]
        L.mi_ref_gear_candidates.restype = C.c_size_t
        L.mi_ref_cdc_select.argtypes = [u64p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32,
                                        u64p, C.c_size_t]
        L.mi_ref_cdc_select.restype = C.c_size_t
        for fn in (L.mi_ref_cdc_two_phase, L.mi_ref_cdc_classic):
            fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(CdcParams), u64p, C.c_size_t]
            fn.restype = C.c_size_t
        L.mi_ref_synth_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.mi_ref_synth_fill.restype = None
        L.mi_ref_synth_fill_many.argtypes = [C.c_uint64, u64p, u64p, u64p, C.c_uint64, C.c_void_p, C.c_int]
        L.mi_ref_synth_fill_many.restype = None
        L.mi_ref_scan_batch.argtypes = [C.c_void_p, u64p, u64p, C.c_uint64, C.POINTER(CdcParams),
                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
        L.mi_ref_scan_batch.restype = C.c_uint64
        L.mi_ref_scan_synthetic.argtypes = [C.c_uint64, u64p, u64p, C.c_uint64, C.POINTER(CdcParams),
                                            C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_uint64, u64p]
        L.mi_ref_scan_synthetic.restype = C.c_uint64
        L.mi_ref_last_phase_seconds.argtypes = [C.POINTER(C.c_double)]
        L.mi_ref_last_phase_seconds.restype = None
        L.mi_ref_dedup_mt.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.mi_ref_dedup_mt.restype = C.c_uint64
        L.mi_ref_chunk_root.argtypes = [C.c_void_p, C.c_uint64, u8p, C.c_int]
        L.mi_ref_chunk_root.restype = None
        L.mi_ref_dedup.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.mi_ref_dedup.restype = C.c_uint64
        L.mi_ref_layer_scan.argtypes = [C.c_void_p, u64p, u64p, C.c_void_p, C.c_uint64, C.c_int, u8p]
        L.mi_ref_layer_scan.restype = C.c_uint64
        L.mi_ref_tar_header.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_char,
                                        C.c_char_p, u8p]
        L.mi_ref_tar_header.restype = C.c_int
        _lib = L
    return _lib


def _buf(data):
    """bytes-like / numpy -> (numpy u8 array kept alive, pointer)"""
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    a = np.ascontiguousarray(a.view(np.uint8))
    return a, a.ctypes.data


def sha256(data, allow_shani=True):
    a, p = _buf(data)
    out = (C.c_uint8 * 32)()
    lib().mi_ref_sha256(p, a.size, out, int(allow_shani))
    return bytes(out)


def have_shani():
    return bool(lib().mi_ref_have_shani())


def crc32(data, crc=0):
    a, p = _buf(data)
    return lib().mi_ref_crc32(crc, p, a.size)


def crc32_combine(c1, c2, len2):
    return lib().mi_ref_crc32_combine(c1, c2, len2)


def gear_table(seed):
    t = np.zeros(256, dtype=np.uint64)
    lib().mi_ref_gear_table(seed, t.ctypes.data_as(C.POINTER(C.c_uint64)))
    return t


def gear_candidates(data, seed, mask_bits, halo=b""):
    a, p = _buf(data)
    h, hp = _buf(halo)
    t = gear_table(seed)
    cap = max(a.size, 1)
    out = np.zeros(cap, dtype=np.uint64)
    n = lib().mi_ref_gear_candidates(p, a.size, hp if h.size else None, h.size,
                                     t.ctypes.data_as(C.POINTER(C.c_uint64)), mask_bits,
                                     out.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
    return out[:n].copy()


def cdc_select(cands, length, min_size, max_size):
    c = np.ascontiguousarray(cands, dtype=np.uint64)
    cap = length // max(min_size, 1) + length // max(max_size, 1) + 2
    out = np.zeros(cap, dtype=np.uint64)
    n = lib().mi_ref_cdc_select(c.ctypes.data_as(C.POINTER(C.c_uint64)), c.size, length, min_size,
                                max_size, out.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
    assert n <= cap
    return out[:n].copy()


def _cdc(fn, data, params):
    a, p = _buf(data)
    cap = a.size // max(params.min_size, 1) + 2
    out = np.zeros(cap, dtype=np.uint64)
    n = fn(p, a.size, C.byref(params), out.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
    assert n <= cap
    return out[:n].copy()


def cdc_two_phase(data, params):
    return _cdc(lib().mi_ref_cdc_two_phase, data, params)


def cdc_classic(data, params):
    return _cdc(lib().mi_ref_cdc_classic, data, params)


def synth_fill(seed, content_id, offset, length):
    out = np.zeros(length, dtype=np.uint8)
    lib().mi_ref_synth_fill(seed, content_id, offset, length, out.ctypes.data)
    return out


def synth_fill_many(seed, content_ids, sizes, n_threads=1):
    """-> (data u8 array, offsets u64 array): the files back to back."""
    szs = np.ascontiguousarray(sizes, dtype=np.uint64)
    offs = np.zeros(szs.size, dtype=np.uint64)
    if szs.size:
        offs[1:] = np.cumsum(szs)[:-1]
    out = np.empty(int(szs.sum()), dtype=np.uint8)
    u64p = C.POINTER(C.c_uint64)
    cp = None
    if content_ids is not None:
        cids = np.ascontiguousarray(content_ids, dtype=np.uint64)
        cp = cids.ctypes.data_as(u64p)
    lib().mi_ref_synth_fill_many(seed, cp, szs.ctypes.data_as(u64p), offs.ctypes.data_as(u64p),
                                 szs.size, out.ctypes.data, n_threads)
    return out, offs


FILE_SHA256, FILE_CRC32, NO_DEDUP = 1, 2, 4


def scan_batch(data, offsets, sizes, params, allow_shani=True, n_threads=1,
               flags=FILE_SHA256 | FILE_CRC32):
    """Returns (files structured array, chunks structured array)."""
    a, p = _buf(data)
    offs = np.ascontiguousarray(offsets, dtype=np.uint64)
    szs = np.ascontiguousarray(sizes, dtype=np.uint64)
    n = offs.size
    cap = int((szs // np.uint64(params.min_size) + np.uint64(2)).sum()) if n else 1
    files = np.zeros(max(n, 1), dtype=FILE_DTYPE)
    chunks = np.zeros(max(cap, 1), dtype=CHUNK_DTYPE)
    u64p = C.POINTER(C.c_uint64)
    total = lib().mi_ref_scan_batch(p, offs.ctypes.data_as(u64p), szs.ctypes.data_as(u64p), n,
                                    C.byref(params), int(allow_shani), n_threads, flags,
                                    files.ctypes.data, chunks.ctypes.data, cap)
    if total == 2**64 - 1:
        raise ValueError("invalid CDC params")
    assert total <= cap
    return files[:n].copy(), chunks[:total].copy()


def scan_synthetic(seed, content_ids, sizes, params, allow_shani=True, n_threads=1, flags=0,
                   chunk_cap=None):
    """scan_batch over synthetic files generated inside the workers (no host copy of the data).
    Returns (files, chunks, n_unique)."""
    szs = np.ascontiguousarray(sizes, dtype=np.uint64)
    n = szs.size
    u64p = C.POINTER(C.c_uint64)
    cp = None
    if content_ids is not None:
        cids = np.ascontiguousarray(content_ids, dtype=np.uint64)
        assert cids.size == n
        cp = cids.ctypes.data_as(u64p)
    cap = int((szs // np.uint64(params.min_size) + np.uint64(2)).sum()) if chunk_cap is None else chunk_cap
    files = np.zeros(max(n, 1), dtype=FILE_DTYPE)
    chunks = np.zeros(max(cap, 1), dtype=CHUNK_DTYPE)
    nu = C.c_uint64()
    total = lib().mi_ref_scan_synthetic(seed, cp, szs.ctypes.data_as(u64p), n, C.byref(params),
                                        int(allow_shani), n_threads, flags, files.ctypes.data,
                                        chunks.ctypes.data, cap, C.byref(nu))
    if total == 2**64 - 1:
        raise ValueError("invalid CDC params")
    assert total <= cap
    return files[:n], chunks[:total], nu.value


def last_phase_seconds():
    out = (C.c_double * 3)()
    lib().mi_ref_last_phase_seconds(out)
    return {"scan_s": out[0], "gather_s": out[1], "dedup_s": out[2]}


def dedup_mt(digests, n_threads):
    d = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros(d.shape[0], dtype=np.int64)
    uniq = lib().mi_ref_dedup_mt(d.ctypes.data, d.shape[0], out.ctypes.data, n_threads)
    return out, uniq


def chunk_root(digests, allow_shani=True):
    d = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 32)
    out = (C.c_uint8 * 32)()
    lib().mi_ref_chunk_root(d.ctypes.data if d.size else None, d.shape[0], out, int(allow_shani))
    return bytes(out)


def dedup(digests):
    d = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros(d.shape[0], dtype=np.int64)
    uniq = lib().mi_ref_dedup(d.ctypes.data, d.shape[0], out.ctypes.data)
    return out, uniq


def layer_scan(data, offsets, sizes, allow_shani=True):
    a, p = _buf(data)
    offs = np.ascontiguousarray(offsets, dtype=np.uint64)
    szs = np.ascontiguousarray(sizes, dtype=np.uint64)
    out = (C.c_uint8 * 32)()
    u64p = C.POINTER(C.c_uint64)
    n = lib().mi_ref_layer_scan(p, offs.ctypes.data_as(u64p), szs.ctypes.data_as(u64p), None,
                                offs.size, int(allow_shani), out)
    return n, bytes(out)


def tar_header(name, size, mtime=0, mode=0o644, typeflag=b"0", linkname=None):
    out = (C.c_uint8 * 512)()
    rc = lib().mi_ref_tar_header(name.encode(), size, mtime, mode, typeflag,
                                 linkname.encode() if linkname else None, out)
    if rc != 0:
        raise ValueError("name too long for ustar")
    return bytes(out)
