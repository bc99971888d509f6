"""makisu_amd -- MI355X (gfx950) layer-snapshot content scan + dedup engine.

Thin ctypes binding over the C ABI (include/makisu_mi.h, libmakisu_mi.so).  It is
the stand-in for the cgo shim a Makisu maintainer would add and
what tests/ and bench.py drive.  No CPU fallback: without the built HIP library
or without a gfx950 device every entry point raises.  Nothing here imports
oracle/.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import build as _build

__all__ = ["Engine", "Batch", "Config", "MiError", "load_library", "FILE_DTYPE", "CHUNK_DTYPE",
           "FLAG_FILE_SHA256", "FLAG_FILE_CRC32", "FLAG_NO_DEDUP", "FLAG_PREFETCH_ROWS", "FLAG_VERIFY_STAGING", "FLAG_FILE_SUMS",
           "SHA_LOADS_AUTO", "SHA_LOADS_LANE", "SHA_LOADS_COOP", "Digest", "digest_hex"]

FLAG_FILE_SHA256 = 0x1
FLAG_FILE_CRC32 = 0x2
FLAG_NO_DEDUP = 0x4
FLAG_PREFETCH_ROWS = 0x8
FLAG_VERIFY_STAGING = 0x10
FLAG_FILE_SUMS = 0x20
SHA_LOADS_AUTO, SHA_LOADS_LANE, SHA_LOADS_COOP = 0, 1, 2
SHA_SCHED_FLAT = 1                       # mi_config.sha_sched: one range, every wave equal


def sha_sched_long_shift(k):
    """MI_SHA_SCHED_LONG_SHIFT(k): the long range of a hashing launch = its first n >> k strings."""
    return ((k & 15) + 1) << 8

ERR_NAMES = {0: "MI_OK", -1: "MI_ERR_INVALID", -2: "MI_ERR_NO_DEVICE", -3: "MI_ERR_HIP",
             -4: "MI_ERR_NOMEM", -5: "MI_ERR_IO", -6: "MI_ERR_STATE", -7: "MI_ERR_CAPACITY"}


class MiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, code), msg))
        self.code = code


class Config(C.Structure):
    """mi_config (include/makisu_mi.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("gear_seed", C.c_uint64),
                ("mask_bits", C.c_uint32), ("min_size", C.c_uint32), ("max_size", C.c_uint32),
                ("flags", C.c_uint32), ("staging_bytes", C.c_uint64), ("n_streams", C.c_uint32),
                ("sha_blocks_per_cu", C.c_uint32), ("sha_load_scheme", C.c_uint32),
                ("sha_coop_min_gib", C.c_uint32), ("sha_coop_blocks_per_cu", C.c_uint32),
                ("sha_sched", C.c_uint32)]


class _FileResult(C.Structure):
    _fields_ = [("user_tag", C.c_uint64), ("size", C.c_uint64), ("first_chunk", C.c_uint64),
                ("n_chunks", C.c_uint32), ("crc32", C.c_uint32), ("chunk_root", C.c_uint8 * 32),
                ("file_sha256", C.c_uint8 * 32)]


class _ChunkResult(C.Structure):
    _fields_ = [("file_index", C.c_uint64), ("offset", C.c_uint64), ("length", C.c_uint32),
                ("reserved", C.c_uint32), ("dup_of", C.c_int64), ("sha256", C.c_uint8 * 32)]


class CtxEntry(C.Structure):
    """mi_ctx_entry: one walked path of a COPY/ADD source tree."""
    _fields_ = [("relpath", C.c_char_p), ("link_target", C.c_char_p), ("file_index", C.c_int64)]


class TreeEntry(C.Structure):
    """mi_tree_entry: one path recorded by mi_batch_add_tree."""
    _fields_ = [("relpath", C.c_char_p), ("link_target", C.c_char_p), ("file_index", C.c_int64),
                ("size", C.c_uint64), ("mtime_sec", C.c_int64), ("mode", C.c_uint32),
                ("kind", C.c_uint8), ("uid", C.c_uint32), ("gid", C.c_uint32)]


TREE_CONTEXT, TREE_SCAN = 0, 1


class SnapshotSide(C.Structure):
    """mi_snapshot_side: one walk (+ optional chunk roots by file_index) for mi_snapshot_diff."""
    _fields_ = [("entries", C.POINTER(TreeEntry)), ("n", C.c_uint64), ("roots", C.c_void_p),
                ("root_stride", C.c_uint64), ("disk_root", C.c_char_p)]


DIFF_SAME, DIFF_CHANGED, DIFF_ANCESTOR = 0, 1, 2


class CopyOp(C.Structure):
    """mi_copy_op."""
    _fields_ = [("src_root", C.c_char_p), ("srcs", C.POINTER(C.c_char_p)), ("n_srcs", C.c_uint64),
                ("dst", C.c_char_p), ("uid", C.c_uint32), ("gid", C.c_uint32)]


class LayerConfig(C.Structure):
    """mi_layer_config."""
    _fields_ = [("struct_size", C.c_uint32), ("gzip_level", C.c_int32), ("out_fd", C.c_int32),
                ("flags", C.c_uint32)]


LAYER_MODE_WITH_TYPE = 0x1
MEMFS_TRUST_CTIME = 0x1


class LayerResult(C.Structure):
    """mi_layer_result: the numbers of step.commitLayer's DigestPair."""
    _fields_ = [("tar_sha256", C.c_uint8 * 32), ("gzip_sha256", C.c_uint8 * 32), ("tar_bytes", C.c_uint64),
                ("gzip_bytes", C.c_uint64), ("n_entries", C.c_uint64)]


GZIP_OFF, GZIP_DEFAULT = -2, -1
# tario.SetCompressionLevel's names (lib/tario/gzip.go:28-43; the `--compression` flag of bin/makisu/cmd/build.go) -> the
# gzip_level of mi_layer_config: "no" is a gzip member of stored blocks (pgzip.NoCompression), not MI_GZIP_OFF
COMPRESSION_LEVELS = {"no": 0, "speed": 1, "size": 9, "default": GZIP_DEFAULT}


def compression_level(name):
    """tario.SetCompressionLevel (lib/tario/gzip.go:36-43): the level for a name, `invalid compression level <name>` otherwise."""
    try:
        return COMPRESSION_LEVELS[name]
    except (KeyError, TypeError):
        raise ValueError("invalid compression level %s" % (name,)) from None


class CommitStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_walked", "n_scanned_files", "scanned_bytes", "n_chunks", "n_layer_entries",
                                          "n_layer_files", "layer_file_bytes", "n_content_changed", "n_roots_learned", "n_content_trusted",
                                          "n_index_new", "n_index_known", "index_new_bytes", "files_opened", "file_bytes_read", "pipelined", "n_windows")] + \
               [(n, C.c_double) for n in ("s_walk_stage", "s_scan", "s_diff", "s_write", "s_total")] + \
               [(n, C.c_uint64) for n in ("n_verified_files", "verified_bytes", "n_refetched", "arena_bytes", "arena_pieces", "arena_moves",
                                          "n_ctxs", "ctx_bytes_max", "ctx_bytes_min", "n_split_files")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PartState(C.Structure):
    """mi_part_state: one part of a split file (include/makisu_mi.h "parts")."""
    _fields_ = [("file_index", C.c_uint64), ("file_size", C.c_uint64), ("begin", C.c_uint64),
                ("end", C.c_uint64), ("entry", C.c_uint64), ("exit", C.c_uint64),
                ("entry_confirmed", C.c_uint32), ("cuts_current", C.c_uint32)]


PART_ALIGN = 262144


class Stats(C.Structure):
    _fields_ = [("bytes_in", C.c_uint64), ("n_files", C.c_uint64), ("n_chunks", C.c_uint64),
                ("n_unique", C.c_uint64), ("ms_h2d", C.c_double), ("ms_cdc", C.c_double),
                ("ms_sort", C.c_double), ("ms_sha_chunks", C.c_double),
                ("ms_sha_files", C.c_double), ("ms_dedup", C.c_double), ("ms_total", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class StageStats(C.Structure):
    """mi_stage_stats."""
    _fields_ = [("spans", C.c_uint64), ("bytes", C.c_uint64), ("verified_spans", C.c_uint64),
                ("mismatches", C.c_uint64), ("repaired", C.c_uint64), ("final_spans", C.c_uint64),
                ("final_mismatches", C.c_uint64), ("ms_verify", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def _np_dtype(struct):
    names, fmts, offs = [], [], []
    for name, ctype in struct._fields_:
        names.append(name)
        offs.append(getattr(struct, name).offset)
        if issubclass(ctype, C.Array):
            fmts.append(("u1", ctype._length_))
        else:
            fmts.append(np.dtype(ctype))
    return np.dtype({"names": names, "formats": fmts, "offsets": offs,
                     "itemsize": C.sizeof(struct)})


FILE_DTYPE = _np_dtype(_FileResult)
CHUNK_DTYPE = _np_dtype(_ChunkResult)

_lib = None


def load_library(rebuild=False):
    """Loads (building first if stale) makisu_amd/libmakisu_mi.so.  Raises if it is missing."""
    global _lib
    if _lib is not None and not rebuild:
        return _lib
    path = os.environ.get("MAKISU_MI_LIB") or _build.LIB    # another build of the library (same-box A/B runs)
    if path == _build.LIB and (rebuild or (_build.needs_build() and os.path.exists(_build.HIPCC))):
        _build.build(force=rebuild)
    if not os.path.exists(path):
        raise MiError(-2, "libmakisu_mi.so is not built (run `python -m makisu_amd.build`); "
                          "there is no CPU fallback")
    L = C.CDLL(path)
    vp, u64, u64p = C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)
    sigs = {
        "mi_abi_version": ([], C.c_int),
        "mi_ctx_warm": ([vp], C.c_int),
        "mi_debug_sha_wave_stats": ([vp, C.c_char_p], C.c_int),
        "mi_config_default": ([C.POINTER(Config)], C.c_int),
        "mi_ctx_create": ([C.POINTER(Config), C.POINTER(vp)], C.c_int),
        "mi_ctx_destroy": ([vp], C.c_int),
        "mi_last_error": ([vp], C.c_char_p),
        "mi_get_stats": ([vp, C.POINTER(Stats)], C.c_int),
        "mi_device_info": ([vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), u64p, C.c_char_p,
                            C.c_size_t], C.c_int),
        "mi_sha_valu_roof": ([vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)], C.c_int),
        "mi_batch_begin": ([vp, u64, u64, C.POINTER(vp)], C.c_int),
        "mi_batch_add_bytes": ([vp, vp, u64, u64], C.c_int),
        "mi_batch_add_path": ([vp, C.c_char_p, u64, u64], C.c_int),
        "mi_batch_reserve": ([vp, u64, u64], C.c_int),
        "mi_batch_add_path_range": ([vp, C.c_char_p, u64, u64, u64], C.c_int),
        "mi_batch_add_paths": ([vp, u64, C.POINTER(C.c_char_p), u64p, u64p], C.c_int),
        "mi_batch_add_synthetic": ([vp, u64, u64p, u64p, u64], C.c_int),
        "mi_batch_add_path_part": ([vp, C.c_char_p, u64, u64, u64, u64], C.c_int),
        "mi_batch_add_synthetic_part": ([vp, u64, u64, u64, u64, u64], C.c_int),
        "mi_chunk_root": ([vp, u64, vp], C.c_int),
        "mi_batch_scan_cuts": ([vp], C.c_int),
        "mi_batch_parts": ([vp, C.POINTER(PartState), u64, u64p], C.c_int),
        "mi_batch_set_part_entry": ([vp, u64, u64], C.c_int),
        "mi_batch_fix_cuts": ([vp], C.c_int),
        "mi_batch_run": ([vp], C.c_int),
        "mi_batch_rerun": ([vp], C.c_int),
        "mi_batch_submit": ([vp], C.c_int),
        "mi_batch_wait": ([vp], C.c_int),
        "mi_batch_counts": ([vp, u64p, u64p, u64p], C.c_int),
        "mi_batch_files": ([vp, vp, u64], C.c_int),
        "mi_batch_chunks": ([vp, vp, u64], C.c_int),
        "mi_batch_chunks_view": ([vp, C.POINTER(vp), u64p], C.c_int),
        "mi_batch_files_view": ([vp, C.POINTER(vp), u64p], C.c_int),
        "mi_batch_device_digests": ([vp, C.POINTER(vp), u64p], C.c_int),
        "mi_batch_read_back": ([vp, vp, u64], C.c_int),
        "mi_batch_reset": ([vp], C.c_int),
        "mi_batch_stage_stats": ([vp, C.POINTER(StageStats)], C.c_int),
        "mi_batch_stage_note": ([vp], C.c_char_p),
        "mi_batch_free": ([vp], C.c_int),
        "mi_dedup_mark": ([vp, vp, u64, vp, u64p], C.c_int),
        "mi_batch_set_global_dedup": ([vp, vp, u64], C.c_int),
        "mi_dedup_mark_range": ([vp, vp, u64, u64, u64, vp, u64p], C.c_int),
        "mi_batch_mark_global": ([vp, vp, u64, u64, u64p], C.c_int),
        "mi_batch_device_dup_of": ([vp, C.POINTER(vp), u64p], C.c_int),
        "mi_sha256_many": ([vp, vp, u64p, u64p, u64, vp], C.c_int),
        "mi_context_checksum": ([vp, vp, u64, C.POINTER(CtxEntry), u64, C.POINTER(C.c_uint32)],
                                C.c_int),
        "mi_batch_add_tree": ([vp, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u64, C.c_uint32,
                               u64p], C.c_int),
        "mi_batch_tree_entries": ([vp, C.POINTER(TreeEntry), u64], C.c_int),
        "mi_context_checksum_tree": ([vp, vp, u64, C.POINTER(C.c_uint32)], C.c_int),
        "mi_tree_walk": ([C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u64, C.c_uint32,
                          C.POINTER(vp), u64p], C.c_int),
        "mi_tree_entries": ([vp, C.POINTER(TreeEntry), u64], C.c_int),
        "mi_tree_free": ([vp], None),
        "mi_entries_commit_order": ([C.POINTER(TreeEntry), u64, u64p], C.c_int),
        "mi_snapshot_diff": ([C.POINTER(SnapshotSide), C.POINTER(SnapshotSide), C.c_int, vp, vp], C.c_int),
        "mi_tar_open": ([C.c_char_p, C.POINTER(vp), u64p], C.c_int),
        "mi_tar_open_ex": ([C.c_char_p, C.POINTER(vp), u64p, C.POINTER(C.c_int), C.c_char_p, u64], C.c_int),
        "mi_tar_inflate": ([C.c_char_p, C.c_char_p, u64p, vp, vp, C.c_char_p, u64], C.c_int),
        "mi_tar_entries": ([vp, C.POINTER(TreeEntry), u64p, u64], C.c_int),
        "mi_tar_free": ([vp], None),
        "mi_entry_similar": ([C.POINTER(TreeEntry), C.POINTER(TreeEntry), C.c_int, vp, vp,
                              C.POINTER(C.c_int)], C.c_int),
        "mi_copy_op_resolve": ([u64, C.c_char_p, C.c_char_p, C.c_char_p, u64, C.c_char_p, u64], C.c_int),
        "mi_resolve_chown": ([C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_char_p, u64], C.c_int),
        "mi_path_match": ([C.c_char_p, C.c_char_p, C.POINTER(C.c_int)], C.c_int),
        "mi_context_sources": ([C.c_char_p, C.POINTER(C.c_char_p), u64, C.c_char_p, u64, u64p, u64p], C.c_int),
        "mi_memfs_create": ([C.c_char_p, C.POINTER(C.c_char_p), u64, C.c_int64, C.POINTER(vp)], C.c_int),
        "mi_memfs_free": ([vp], None),
        "mi_memfs_error": ([vp], C.c_char_p),
        "mi_memfs_set_clock": ([vp, C.c_int64], C.c_int),
        "mi_memfs_reset": ([vp], C.c_int),
        "mi_memfs_update_from_entries": ([vp, C.POINTER(TreeEntry), u64, u64p], C.c_int),
        "mi_memfs_untar": ([vp, C.c_char_p, C.POINTER(TreeEntry), u64p, u64, u64p], C.c_int),
        "mi_memfs_add_layer_by_scan": ([vp, C.POINTER(TreeEntry), u64, vp, u64, C.POINTER(vp), u64p], C.c_int),
        "mi_memfs_add_layer_by_copy_ops": ([vp, C.POINTER(CopyOp), u64, C.POINTER(vp), u64p], C.c_int),
        "mi_memfs_entries": ([vp, C.POINTER(TreeEntry), C.POINTER(C.c_char_p), u64, u64p], C.c_int),
        "mi_memfs_checkpoint": ([vp, C.c_char_p, C.POINTER(C.c_char_p), u64], C.c_int),
        "mi_memfs_commit_layer": ([vp, vp, C.c_int, C.POINTER(CopyOp), u64, C.POINTER(LayerConfig), C.POINTER(LayerResult),
                                   C.POINTER(vp), C.POINTER(C.c_int)], C.c_int),
        "mi_memfs_commit_layer_n": ([vp, C.POINTER(vp), C.c_uint32, C.c_int, C.POINTER(CopyOp), u64, C.POINTER(LayerConfig),
                                     C.POINTER(LayerResult), C.POINTER(vp), C.POINTER(C.c_int)], C.c_int),
        "mi_memfs_commit_stats": ([vp, C.POINTER(CommitStats)], C.c_int),
        "mi_memfs_set_index": ([vp, vp], C.c_int),
        "mi_memfs_set_options": ([vp, C.c_uint32], C.c_int),
        "mi_memfs_release_device": ([vp], C.c_int),
        "mi_memfs_reserve_device": ([vp, vp, u64, u64], C.c_int),
        "mi_memfs_root_of": ([vp, C.c_char_p, vp, C.POINTER(C.c_int)], C.c_int),
        "mi_copy_layer_roots": ([vp, vp, vp, u64], C.c_int),
        "mi_batch_roots": ([vp, vp, u64], C.c_int),
        "mi_batch_read_file": ([vp, u64, u64, vp, u64], C.c_int),
        "mi_layer_add_batch_file": ([vp, C.POINTER(TreeEntry), vp, u64], C.c_int),
        "mi_layer_io_counts": ([vp, u64p, u64p], C.c_int),
        "mi_copy_op_execute": ([C.POINTER(CopyOp), C.c_uint32, C.POINTER(C.c_char_p), u64, C.c_char_p, u64], C.c_int),
        "mi_copy_layer_entries": ([vp, C.POINTER(TreeEntry), C.POINTER(C.c_char_p), u64], C.c_int),
        "mi_copy_layer_free": ([vp], None),
        "mi_layer_config_default": ([C.POINTER(LayerConfig)], C.c_int),
        "mi_layer_begin": ([C.POINTER(LayerConfig), C.POINTER(vp)], C.c_int),
        "mi_layer_add": ([vp, C.POINTER(TreeEntry), C.c_char_p], C.c_int),
        "mi_layer_add_whiteout": ([vp, C.c_char_p], C.c_int),
        "mi_layer_finish": ([vp, C.POINTER(LayerResult)], C.c_int),
        "mi_layer_error": ([vp], C.c_char_p),
        "mi_layer_free": ([vp], None),
        "mi_layer_header_bytes": ([C.POINTER(TreeEntry), C.c_uint32, vp, u64, u64p], C.c_int),
        "mi_cache_parse_entry_str": ([C.c_char_p, C.c_char_p, u64, C.c_char_p, u64], C.c_int),
        "mi_cache_key": ([C.c_char_p, C.c_char_p, u64], C.c_int),
        "mi_cache_create_entry": ([vp, vp, C.c_char_p, u64], C.c_int),
        "mi_cache_parse_entry": ([C.c_char_p, C.POINTER(C.c_int), vp, vp], C.c_int),
        "mi_comm_unique_id": ([vp], C.c_int),
        "mi_comm_init_rank": ([vp, C.c_int, C.c_int, vp], C.c_int),
        "mi_comm_init_all": ([C.POINTER(vp), C.c_int], C.c_int),
        "mi_comm_destroy": ([vp], C.c_int),
        "mi_comm_ranks": ([vp, C.POINTER(C.c_int)], C.c_int),
        "mi_comm_exchange_ms": ([vp, C.POINTER(C.c_double), C.POINTER(C.c_double)], C.c_int),
        "mi_dedup_allgather": ([vp, u64p, u64p, u64p], C.c_int),
        "mi_dedup_allgather_all": ([C.POINTER(vp), C.c_int, u64p, u64p], C.c_int),
        "mi_dedup_alltoall": ([vp, u64p, u64p, u64p], C.c_int),
        "mi_dedup_alltoall_all": ([C.POINTER(vp), C.c_int, u64p, u64p], C.c_int),
        "mi_index_create": ([vp, u64, C.POINTER(vp)], C.c_int),
        "mi_index_free": ([vp], None),
        "mi_index_count": ([vp, u64p], C.c_int),
        "mi_index_add_batch": ([vp, vp, vp, u64, u64p, u64p], C.c_int),
        "mi_index_export": ([vp, vp, u64], C.c_int),
        "mi_index_import": ([vp, vp, u64, u64p], C.c_int),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(L, name)          # AttributeError here = header/library drift
        fn.argtypes = args
        fn.restype = res
    L._mi_symbols = tuple(sigs)
    _lib = L
    return L


def digest_hex(raw32):
    return bytes(bytearray(raw32)).hex()


class Digest(str):
    """"sha256:<hex>" -- mirrors image.Digest (lib/docker/image/digest.go:26-50)."""

    @classmethod
    def from_raw(cls, raw32):
        return cls("sha256:" + digest_hex(raw32))

    def hex(self):                      # Digest.Hex()
        return self[self.index(":") + 1:]


KIND_DIR, KIND_FILE, KIND_SYMLINK, KIND_HARDLINK = 0, 1, 2, 3


def entry_similar(a, b, ignore_time=False, root_a=None, root_b=None):
    """tario.IsSimilarHeader on two entry dicts (keys: relpath, link_target, size, mtime_sec, mode,
    kind, uid, gid; missing keys = 0/None), optionally content-aware through 32-byte chunk roots."""
    def make(d):
        e = TreeEntry()
        e.relpath = os.fsencode(d.get("relpath", "x"))
        lt = d.get("link_target")
        e.link_target = os.fsencode(lt) if lt is not None else None
        e.file_index = d.get("file_index", -1)
        for k in ("size", "mtime_sec", "mode", "kind", "uid", "gid"):
            setattr(e, k, d.get(k, 0))
        return e
    ea, eb, out = make(a), make(b), C.c_int()
    ra = (C.c_uint8 * 32).from_buffer_copy(bytes(root_a)) if root_a is not None else None
    rb = (C.c_uint8 * 32).from_buffer_copy(bytes(root_b)) if root_b is not None else None
    rc = load_library().mi_entry_similar(C.byref(ea), C.byref(eb), int(ignore_time), ra, rb, C.byref(out))
    if rc:
        raise MiError(rc, "mi_entry_similar: unsupported type")
    return bool(out.value)


def commit_order(relpaths):
    """memLayer.rangeFiles' order (sort.Strings over absolute dst paths, whiteout markers under the
    path they delete) for a list of relpaths: list of indices in commit order."""
    n = len(relpaths)
    arr = (TreeEntry * max(n, 1))()
    keep = [os.fsencode(r) for r in relpaths]
    for i, r in enumerate(keep):
        arr[i].relpath = r
    out = (C.c_uint64 * max(n, 1))()
    rc = load_library().mi_entries_commit_order(arr, n, out)
    if rc:
        raise MiError(rc, "mi_entries_commit_order")
    return [int(out[i]) for i in range(n)]


def tar_entries(path):
    """mi_tar_open + mi_tar_entries: the entries of an uncompressed layer tar as dicts (the keys of
    tree_walk(full=True) plus "data_offset": where a regular file's bytes start in the archive)."""
    L = load_library()
    h, n = C.c_void_p(), C.c_uint64()
    err = C.create_string_buffer(512)
    rc = L.mi_tar_open_ex(os.fsencode(path), C.byref(h), C.byref(n), None, err, len(err))
    if rc:
        raise MiError(rc, "mi_tar_open: %s" % err.value.decode(errors="replace"))
    try:
        arr = (TreeEntry * max(n.value, 1))()
        offs = (C.c_uint64 * max(n.value, 1))()
        rc = L.mi_tar_entries(h, arr, offs, n.value)
        if rc:
            raise MiError(rc, "mi_tar_entries")
        out = []
        for i in range(n.value):
            d = _entry_dict(arr[i])
            d["data_offset"] = int(offs[i])
            out.append(d)
        return out
    finally:
        L.mi_tar_free(h)


def tar_inflate(blob_path, tar_path_out=None):
    """mi_tar_inflate: a stored (gzip) layer blob -> uncompressed tar file; returns
    dict(tar_bytes, tar_digest, blob_digest)."""
    nb = C.c_uint64()
    t, g = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
    err = C.create_string_buffer(512)
    rc = load_library().mi_tar_inflate(os.fsencode(blob_path),
                                       os.fsencode(tar_path_out) if tar_path_out is not None else None,
                                       C.byref(nb), t, g, err, len(err))
    if rc:
        raise MiError(rc, "mi_tar_inflate: %s" % err.value.decode(errors="replace"))
    return {"tar_bytes": nb.value, "tar_digest": Digest.from_raw(t), "blob_digest": Digest.from_raw(g)}


def _entry_array(dicts, keep):
    arr = (TreeEntry * max(len(dicts), 1))()
    for i, d in enumerate(dicts):
        rp = os.fsencode(d.get("relpath", ""))
        lt = d.get("link_target")
        lt = os.fsencode(lt) if lt is not None else None
        keep += [rp, lt]
        arr[i].relpath, arr[i].link_target = rp, lt
        arr[i].file_index = d.get("file_index", -1)
        for k in ("size", "mtime_sec", "mode", "kind", "uid", "gid"):
            setattr(arr[i], k, d.get(k, 0))
    return arr


def apply_layer(base, layer, root=None, blacklist=()):
    """Harness helper over the MemFS handle (the library has ONE layer merge, mi_memfs_update_from_entries):
    UpdateFromTarReader of `base`, then of `layer`, into a fresh tree; returns mi_memfs_entries -- the merged tree in
    sorted-path order as entry dicts (relpaths without the leading "/", a "src" key; the directories addAncestors
    created are entries like any other).  root = the directory the layers are (or would be) untarred to: its
    blacklist / mountpoint filter applies; None = an empty scratch directory, where nothing is mounted or
    blacklisted (special files and ".wh..wh." metadata are skipped either way, as in the reference)."""
    import tempfile
    scratch = tempfile.mkdtemp(prefix="mi_memfs_") if root is None else None
    try:
        with MemFS(root if root is not None else scratch, blacklist) as fs:
            if base:
                fs.update_from_entries(base)
            fs.update_from_entries(layer)
            return fs.entries()
    finally:
        if scratch:
            os.rmdir(scratch)


def snapshot_diff(before, after, ignore_time=False, roots_before=None, roots_after=None, disk_root=None):
    """mi_snapshot_diff on two lists of entry dicts (tree_walk(..., full=True)); roots_* = (n_files,
    32) uint8 arrays indexed by file_index, or None.  Returns (flags per `after` entry, whiteout
    flags per `before` entry) as lists of ints."""
    keep = []
    sides = []
    for ents, roots in ((before, roots_before), (after, roots_after)):
        side = SnapshotSide()
        arr = _entry_array(ents, keep)
        keep.append(arr)
        side.entries, side.n = arr, len(ents)
        if roots is not None:
            r = np.ascontiguousarray(roots, dtype=np.uint8).reshape(-1, 32)
            keep.append(r)
            side.roots, side.root_stride = r.ctypes.data, 32
        sides.append(side)
    if disk_root is not None:
        sides[1].disk_root = os.fsencode(disk_root)
    flags = np.zeros(max(len(after), 1), dtype=np.uint8)
    wh = np.zeros(max(len(before), 1), dtype=np.uint8)
    rc = load_library().mi_snapshot_diff(C.byref(sides[0]), C.byref(sides[1]), int(ignore_time),
                                         flags.ctypes.data, wh.ctypes.data)
    if rc:
        raise MiError(rc, "mi_snapshot_diff")
    return flags[: len(after)].tolist(), wh[: len(before)].tolist()


def _entry_dict(e):
    return {"relpath": os.fsdecode(e.relpath), "link_target": os.fsdecode(e.link_target) if e.link_target else None,
            "file_index": e.file_index, "size": e.size, "mtime_sec": e.mtime_sec, "mode": e.mode,
            "kind": e.kind, "uid": e.uid, "gid": e.gid}


def tree_walk(root, rel_base=None, blacklist=(), mode=TREE_CONTEXT, full=False):
    """The reference's directory walk on its own (host logic, no GPU): list of
    (relpath, link_target, file_ordinal, size, kind, mode) in visit order (full=True: dicts
    with every mi_tree_entry field)."""
    L = load_library()
    bl = (C.c_char_p * max(len(blacklist), 1))(*[os.fsencode(x) for x in blacklist])
    h, n = C.c_void_p(), C.c_uint64()
    rc = L.mi_tree_walk(os.fsencode(root), os.fsencode(rel_base) if rel_base else None, bl,
                        len(blacklist), mode, C.byref(h), C.byref(n))
    if rc:
        raise MiError(rc, "mi_tree_walk(%s)" % root)
    try:
        arr = (TreeEntry * max(n.value, 1))()
        rc = L.mi_tree_entries(h, arr, n.value)
        if rc:
            raise MiError(rc, "mi_tree_entries")
        if full:
            return [_entry_dict(e) for e in arr[:n.value]]
        return [(os.fsdecode(e.relpath), os.fsdecode(e.link_target) if e.link_target else None,
                 e.file_index, e.size, e.kind, e.mode) for e in arr[:n.value]]
    finally:
        L.mi_tree_free(h)


def copy_op_resolve(n_srcs, work_dir, dst):
    """mi_copy_op_resolve: NewCopyOperation's checks + resolveDestination; returns the absolute dst."""
    out = C.create_string_buffer(4096)
    err = C.create_string_buffer(300)
    rc = load_library().mi_copy_op_resolve(n_srcs, os.fsencode(work_dir) if work_dir is not None else None, os.fsencode(dst), out,
                              len(out), err, len(err))
    if rc:
        raise MiError(rc, err.value.decode(errors="replace"))
    return os.fsdecode(out.value)


def resolve_chown(chown, preserve_owner=False):
    """mi_resolve_chown: (uid, gid) of a --chown string, the way utils.ResolveChown reads it."""
    uid, gid = C.c_int64(), C.c_int64()
    err = C.create_string_buffer(300)
    rc = load_library().mi_resolve_chown(os.fsencode(chown), int(preserve_owner), C.byref(uid), C.byref(gid), err, len(err))
    if rc:
        raise MiError(rc, err.value.decode(errors="replace"))
    return uid.value, gid.value


def path_match(pattern, name):
    """mi_path_match: Go's filepath.Match; raises MiError(MI_ERR_INVALID) for ErrBadPattern."""
    m = C.c_int()
    rc = load_library().mi_path_match(os.fsencode(pattern), os.fsencode(name), C.byref(m))
    if rc:
        raise MiError(rc, "syntax error in pattern")
    return bool(m.value)


def context_sources(context_root, from_paths):
    """mi_context_sources: addCopyStep.resolveFromPaths -- the sources joined to the context root and globbed."""
    arr = (C.c_char_p * max(len(from_paths), 1))(*[os.fsencode(x) for x in from_paths])
    n, nbytes = C.c_uint64(), C.c_uint64()
    L = load_library()
    rc = L.mi_context_sources(os.fsencode(context_root), arr, len(from_paths), None, 0, C.byref(n), C.byref(nbytes))
    if rc not in (0, -7):
        raise MiError(rc, "mi_context_sources")
    buf = C.create_string_buffer(max(nbytes.value, 1))
    rc = L.mi_context_sources(os.fsencode(context_root), arr, len(from_paths), buf, nbytes.value, C.byref(n), C.byref(nbytes))
    if rc:
        raise MiError(rc, "mi_context_sources")
    return [os.fsdecode(x) for x in buf.raw[:nbytes.value].split(b"\0")[:n.value]]


def _copy_op_array(ops, keep):
    cops = (CopyOp * max(len(ops), 1))()
    for i, o in enumerate(ops):
        srcs = [os.fsencode(x) for x in o["srcs"]]
        sarr = (C.c_char_p * max(len(srcs), 1))(*srcs)
        keep += [srcs, sarr]
        cops[i].src_root = os.fsencode(o["src_root"])
        cops[i].srcs, cops[i].n_srcs = sarr, len(srcs)
        cops[i].dst = os.fsencode(o["dst"])
        cops[i].uid, cops[i].gid = o.get("uid", 0), o.get("gid", 0)
    return cops


def _take_copy_layer(L, h, n):
    """mi_copy_layer -> list of entry dicts (commit order) with an extra "src" key and, for regular files of a
    content-aware commit, "root" (32 bytes); frees the handle."""
    try:
        out = (TreeEntry * max(n, 1))()
        srcp = (C.c_char_p * max(n, 1))()
        rc = L.mi_copy_layer_entries(h, out, srcp, n)
        if rc:
            raise MiError(rc, "mi_copy_layer_entries")
        roots = np.zeros((max(n, 1), 32), dtype=np.uint8)
        has = np.zeros(max(n, 1), dtype=np.uint8)
        rc = L.mi_copy_layer_roots(h, roots.ctypes.data, has.ctypes.data, n)
        if rc:
            raise MiError(rc, "mi_copy_layer_roots")
        res = []
        for i in range(n):
            d = _entry_dict(out[i])
            d["src"] = os.fsdecode(srcp[i]) if srcp[i] is not None else ""
            if has[i]:
                d["root"] = roots[i].tobytes()
            res.append(d)
        return res
    finally:
        L.mi_copy_layer_free(h)


COPY_CHOWN, COPY_INTERNAL, COPY_PRESERVE_OWNER = 1, 2, 4


def copy_op_execute(op, chown=False, internal=False, preserve_owner=False, blacklist=()):
    """mi_copy_op_execute: CopyOperation.Execute -- the on-disk copy of a COPY/ADD step."""
    keep = []
    cops = _copy_op_array([op], keep)
    bl = (C.c_char_p * max(len(blacklist), 1))(*[os.fsencode(x) for x in blacklist])
    err = C.create_string_buffer(600)
    flags = (COPY_CHOWN if chown else 0) | (COPY_INTERNAL if internal else 0) | (COPY_PRESERVE_OWNER if preserve_owner else 0)
    rc = load_library().mi_copy_op_execute(cops, flags, bl, len(blacklist), err, len(err))
    if rc:
        raise MiError(rc, "mi_copy_op_execute: %s" % err.value.decode(errors="replace"))


def copy_ops_layer(tree, tree_root, ops, now_sec=0):
    """Harness helper over the MemFS handle: a tree rooted at tree_root (an existing directory) that holds `tree`
    (entry dicts, merged as a base layer would be), then AddLayerByCopyOps(ops) -- ops = dicts(src_root, srcs, dst,
    uid, gid).  Returns the layer as entry dicts in commit order with an extra "src" key."""
    with MemFS(tree_root, now_sec=now_sec) as fs:
        if tree:
            fs.update_from_entries(tree)
        return fs.add_layer_by_copy_ops(ops)


class MemFS:
    """mi_memfs_*: the reference's MemFS (lib/snapshot/mem_fs.go) as a handle -- one tree for the life of a build.

    fs = MemFS(root, blacklist); fs.update_from_entries(tar_entries); layer = fs.add_layer_by_scan(walk_entries);
    layer = fs.add_layer_by_copy_ops([op, ...]); fs.entries()."""

    def __init__(self, root, blacklist=(), now_sec=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        bl = (C.c_char_p * max(len(blacklist), 1))(*[os.fsencode(x) for x in blacklist])
        rc = self._lib.mi_memfs_create(os.fsencode(root), bl, len(blacklist), now_sec, C.byref(self._h))
        if rc:
            raise MiError(rc, "mi_memfs_create: unable to stat root dir: %s" % root)
        self.root, self.blacklist = root, list(blacklist)

    def _check(self, rc, what):
        if rc:
            raise MiError(rc, "%s: %s" % (what, self._lib.mi_memfs_error(self._h).decode(errors="replace")))

    def close(self):
        if self._h:
            self._lib.mi_memfs_free(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_clock(self, now_sec):
        self._check(self._lib.mi_memfs_set_clock(self._h, now_sec), "mi_memfs_set_clock")

    def reset(self):
        self._check(self._lib.mi_memfs_reset(self._h), "mi_memfs_reset")

    def update_from_entries(self, layer):
        """UpdateFromTarReader (untar = false); returns the number of headers merged."""
        keep = []
        arr = _entry_array(layer, keep)
        n = C.c_uint64()
        self._check(self._lib.mi_memfs_update_from_entries(self._h, arr, len(layer), C.byref(n)), "mi_memfs_update_from_entries")
        return n.value

    def update_from_tar(self, tar_path, untar=False):
        """UpdateFromTarPath on a PLAIN tar: its headers merged into the tree; untar=True also writes it below the root
        (untarOneItem).  Returns the number of headers merged."""
        ents = tar_entries(tar_path)
        if not untar:
            return self.update_from_entries(ents)
        keep = []
        arr = _entry_array(ents, keep)
        offs = (C.c_uint64 * max(len(ents), 1))(*[e["data_offset"] for e in ents])
        n = C.c_uint64()
        self._check(self._lib.mi_memfs_untar(self._h, os.fsencode(tar_path), arr, offs, len(ents), C.byref(n)), "mi_memfs_untar")
        return n.value

    def add_layer_by_scan(self, walked, roots=None):
        """createLayerByScan on a walk of the root (tree_walk(root, root, blacklist, TREE_SCAN, full=True)); roots: an
        (n_files, 32) uint8 array indexed by file_index, or None."""
        keep = []
        arr = _entry_array(walked, keep)
        h, n = C.c_void_p(), C.c_uint64()
        rp, stride = None, 0
        if roots is not None:
            roots = np.ascontiguousarray(roots, dtype=np.uint8)
            rp, stride = roots.ctypes.data, roots.strides[0] if roots.ndim == 2 else 32
        self._check(self._lib.mi_memfs_add_layer_by_scan(self._h, arr, len(walked), rp, stride, C.byref(h), C.byref(n)),
                    "mi_memfs_add_layer_by_scan")
        return _take_copy_layer(self._lib, h, n.value)

    def add_layer_by_copy_ops(self, ops):
        keep = []
        cops = _copy_op_array(ops, keep)
        h, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.mi_memfs_add_layer_by_copy_ops(self._h, cops, len(ops), C.byref(h), C.byref(n)),
                    "mi_memfs_add_layer_by_copy_ops")
        return _take_copy_layer(self._lib, h, n.value)

    def commit_layer(self, must_scan=False, ops=(), out_fd=-1, gzip_level=GZIP_DEFAULT, engine=None, mode_with_type=False, want_layer=True):
        """step.commitLayer: the layer by scan or by copy ops, written through the layer writer.  engine = an Engine: the
        content-aware commit (walk + stage + GPU scan + diff with chunk roots + tar from HBM, one call); a list of Engines: the
        same over several GPUs (mi_memfs_commit_layer_n); None: the reference's.  Returns None when there is nothing to do, else
        dict(tar_digest, gzip_digest, tar_bytes, gzip_bytes, n_entries, layer=[entries, with "root" for scanned files], stats={...});
        want_layer=False leaves the layer's entries in the library (layer=None): a timing harness should not measure 100 000 dicts."""
        keep = []
        cops = _copy_op_array(list(ops), keep)
        cfg = LayerConfig()
        self._lib.mi_layer_config_default(C.byref(cfg))
        cfg.out_fd, cfg.gzip_level = out_fd, gzip_level
        if mode_with_type:
            cfg.flags |= LAYER_MODE_WITH_TYPE
        res, h, done = LayerResult(), C.c_void_p(), C.c_int()
        hp = C.byref(h) if want_layer else None
        if isinstance(engine, (list, tuple)):
            for e in engine:
                e._children.add(self)
            ctxs = (C.c_void_p * max(len(engine), 1))(*[e._h for e in engine])
            self._check(self._lib.mi_memfs_commit_layer_n(self._h, ctxs, len(engine), int(must_scan), cops, len(ops), C.byref(cfg),
                                                          C.byref(res), hp, C.byref(done)), "mi_memfs_commit_layer_n")
        else:
            ctx = engine._h if engine is not None else None
            if engine is not None:
                engine._children.add(self)                    # the handle keeps a batch of that ctx: given back before it dies
            self._check(self._lib.mi_memfs_commit_layer(self._h, ctx, int(must_scan), cops, len(ops), C.byref(cfg), C.byref(res),
                                                        hp, C.byref(done)), "mi_memfs_commit_layer")
        if not done.value:
            return None
        return {"tar_digest": Digest.from_raw(res.tar_sha256), "gzip_digest": Digest.from_raw(res.gzip_sha256),
                "tar_bytes": res.tar_bytes, "gzip_bytes": res.gzip_bytes, "n_entries": res.n_entries,
                "layer": _take_copy_layer(self._lib, h, int(res.n_entries)) if want_layer else None, "stats": self.commit_stats()}

    def commit_stats(self):
        st = CommitStats()
        self._check(self._lib.mi_memfs_commit_stats(self._h, C.byref(st)), "mi_memfs_commit_stats")
        return st.as_dict()

    def set_index(self, index):
        """every content-aware commit adds its batch's chunks to `index` (a ChunkIndex of the same Engine); None stops"""
        self._index = index                                   # (kept alive)
        self._check(self._lib.mi_memfs_set_index(self._h, index._h if index is not None else None), "mi_memfs_set_index")

    def reserve_device(self, engine, files, nbytes):
        """the handle's batch ahead of its first commit (a ctx's first use costs: see the header)"""
        engine._children.add(self)
        self._check(self._lib.mi_memfs_reserve_device(self._h, engine._h, files, nbytes), "mi_memfs_reserve_device")

    def set_options(self, trust_ctime=False):
        """MI_MEMFS_TRUST_CTIME: scan commits do not read files again whose inode is what it was when they were hashed"""
        self._check(self._lib.mi_memfs_set_options(self._h, MEMFS_TRUST_CTIME if trust_ctime else 0), "mi_memfs_set_options")

    def release_device(self):
        if self._h:
            self._check(self._lib.mi_memfs_release_device(self._h), "mi_memfs_release_device")

    free = release_device                                     # what Engine.close asks of its children

    def root_of(self, path):
        """the chunk root the tree holds for a path ("/"-rooted below the root), or None if it was never scanned"""
        out = (C.c_uint8 * 32)()
        has = C.c_int()
        self._check(self._lib.mi_memfs_root_of(self._h, os.fsencode(path), out, C.byref(has)), "mi_memfs_root_of")
        return bytes(out) if has.value else None

    def checkpoint(self, new_root, sources):
        """MemFS.Checkpoint: copy what a later stage will COPY --from below new_root."""
        arr = (C.c_char_p * max(len(sources), 1))(*[os.fsencode(x) for x in sources])
        self._check(self._lib.mi_memfs_checkpoint(self._h, os.fsencode(new_root), arr, len(sources)), "mi_memfs_checkpoint")

    def scan(self):
        """walk the root with this MemFS's blacklist, then add_layer_by_scan"""
        return self.add_layer_by_scan(tree_walk(self.root, self.root, self.blacklist, TREE_SCAN, full=True))

    def entries(self):
        n = C.c_uint64()
        rc = self._lib.mi_memfs_entries(self._h, None, None, 0, C.byref(n))
        if rc not in (0, -7):
            self._check(rc, "mi_memfs_entries")
        out = (TreeEntry * max(n.value, 1))()
        srcp = (C.c_char_p * max(n.value, 1))()
        self._check(self._lib.mi_memfs_entries(self._h, out, srcp, n.value, C.byref(n)), "mi_memfs_entries")
        res = []
        for i in range(n.value):
            d = _entry_dict(out[i])
            d["src"] = os.fsdecode(srcp[i]) if srcp[i] is not None else ""
            res.append(d)
        return res


class Layer:
    """mi_layer_*: the layer tar writer with its two stream digests (host threads, no GPU).

    with Layer(out_fd=fd, gzip_level=GZIP_DEFAULT) as l:
        l.add(entry_dict, src_path); l.add_whiteout("/deleted/path"); pair = l.finish()
    finish() -> dict(tar_digest="sha256:..", gzip_digest=.., tar_bytes, gzip_bytes, n_entries)."""

    def __init__(self, out_fd=-1, gzip_level=GZIP_DEFAULT, mode_with_type=False):
        self._lib = load_library()
        cfg = LayerConfig()
        self._lib.mi_layer_config_default(C.byref(cfg))
        cfg.out_fd, cfg.gzip_level = out_fd, gzip_level
        cfg.flags = LAYER_MODE_WITH_TYPE if mode_with_type else 0
        self._gzip = gzip_level != GZIP_OFF
        self._h = C.c_void_p()
        rc = self._lib.mi_layer_begin(C.byref(cfg), C.byref(self._h))
        if rc:
            raise MiError(rc, "mi_layer_begin")

    def _check(self, rc):
        if rc:
            raise MiError(rc, self._lib.mi_layer_error(self._h).decode(errors="replace"))

    def add(self, entry, src_path=None):
        keep = []
        arr = _entry_array([entry], keep)
        self._check(self._lib.mi_layer_add(self._h, arr, os.fsencode(src_path) if src_path is not None else None))

    def add_batch_file(self, entry, batch, file_index):
        """the entry with its content taken from a staged batch (the bytes the GPU scanned), not from a path"""
        keep = []
        arr = _entry_array([entry], keep)
        self._check(self._lib.mi_layer_add_batch_file(self._h, arr, batch._h, file_index))

    def io_counts(self):
        o, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.mi_layer_io_counts(self._h, C.byref(o), C.byref(b)))
        return o.value, b.value

    def add_whiteout(self, deleted_path):
        self._check(self._lib.mi_layer_add_whiteout(self._h, os.fsencode(deleted_path)))

    def finish(self):
        res = LayerResult()
        self._check(self._lib.mi_layer_finish(self._h, C.byref(res)))
        return {"tar_digest": Digest.from_raw(res.tar_sha256),
                "gzip_digest": Digest.from_raw(res.gzip_sha256) if self._gzip else None,
                "tar_sha256": bytes(res.tar_sha256), "gzip_sha256": bytes(res.gzip_sha256),
                "tar_bytes": res.tar_bytes, "gzip_bytes": res.gzip_bytes, "n_entries": res.n_entries}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi_layer_free(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def layer_header_bytes(entry, mode_with_type=False):
    """The tar header block(s) the layer writer emits for one entry dict (mode_with_type: the Mode field
    keeps the st_mode's file-type bits like Go <= 1.8 wrote them, MI_LAYER_MODE_WITH_TYPE)."""
    keep = []
    arr = _entry_array([entry], keep)
    n = C.c_uint64()
    buf = (C.c_uint8 * 8192)()
    rc = load_library().mi_layer_header_bytes(arr, LAYER_MODE_WITH_TYPE if mode_with_type else 0, buf, 8192, C.byref(n))
    if rc:
        raise MiError(rc, "mi_layer_header_bytes")
    return bytes(buf[: n.value])


def chunk_root(digests):
    """mi_chunk_root: the file-level root of a list of chunk digests (bytes-like of n x 32, or an
    (n, 32) uint8 array) -- what a split file's parts combine to."""
    a = np.ascontiguousarray(np.frombuffer(digests, dtype=np.uint8) if not isinstance(digests, np.ndarray) else digests,
                             dtype=np.uint8).reshape(-1, 32)
    out = np.zeros(32, dtype=np.uint8)
    rc = load_library().mi_chunk_root(a.ctypes.data if a.size else None, a.shape[0], out.ctypes.data)
    if rc:
        raise MiError(rc, "mi_chunk_root")
    return out.tobytes()


def cache_key(cache_id):
    out = C.create_string_buffer(len(os.fsencode(cache_id)) + 64)
    rc = load_library().mi_cache_key(os.fsencode(cache_id), out, len(out))
    if rc:
        raise MiError(rc, "mi_cache_key")
    return out.value.decode()


def cache_create_entry(tar_sha256=None, gzip_sha256=None):
    """createEntry: raw 32-byte digests -> "tarHex,gzipHex"; (None, None) -> the empty marker."""
    out = C.create_string_buffer(160)
    t = (C.c_uint8 * 32).from_buffer_copy(tar_sha256) if tar_sha256 is not None else None
    g = (C.c_uint8 * 32).from_buffer_copy(gzip_sha256) if gzip_sha256 is not None else None
    rc = load_library().mi_cache_create_entry(t, g, out, len(out))
    if rc:
        raise MiError(rc, "mi_cache_create_entry")
    return out.value.decode()


def cache_parse_entry(entry):
    """parseEntry: -> None for the empty marker, else (tar Digest, gzip Digest); ValueError if malformed."""
    t, g, e = (C.c_uint8 * 32)(), (C.c_uint8 * 32)(), C.c_int()
    rc = load_library().mi_cache_parse_entry(entry.encode(), C.byref(e), t, g)
    if rc:
        raise ValueError("parse redis entry: %s" % entry)
    if e.value:
        return None
    return Digest.from_raw(t), Digest.from_raw(g)


def cache_parse_entry_str(entry):
    """parseEntry to the letter: ("sha256:<first half>", "sha256:<rest>"); ValueError without a comma."""
    raw = entry.encode()
    a, b = C.create_string_buffer(len(raw) + 16), C.create_string_buffer(len(raw) + 16)
    rc = load_library().mi_cache_parse_entry_str(raw, a, len(a), b, len(b))
    if rc:
        raise ValueError("parse redis entry: %s" % entry)
    return Digest(a.value.decode()), Digest(b.value.decode())


class ChunkIndex:
    """mi_index_*: device-resident digest set; add_batch() says which chunks earlier
    batches already held, export()/load() move it as bytes (keyvalue.Store value)."""

    def __init__(self, engine, capacity_hint=0):
        self._eng = engine
        self._lib = engine._lib
        self._h = C.c_void_p()
        engine._check(self._lib.mi_index_create(engine._h, capacity_hint, C.byref(self._h)))
        engine._children.add(self)

    def __len__(self):
        n = C.c_uint64()
        self._eng._check(self._lib.mi_index_count(self._h, C.byref(n)))
        return n.value

    def add_batch(self, batch):
        """-> (known flags per chunk as np.uint8, n_new, n_known)."""
        n = batch.counts()[1]
        known = np.zeros(max(n, 1), dtype=np.uint8)
        n_new, n_known = C.c_uint64(), C.c_uint64()
        self._eng._check(self._lib.mi_index_add_batch(self._h, batch._h, known.ctypes.data, n,
                                                      C.byref(n_new), C.byref(n_known)))
        return known[:n], n_new.value, n_known.value

    def export(self):
        n = len(self)
        out = np.zeros((max(n, 1), 32), dtype=np.uint8)
        self._eng._check(self._lib.mi_index_export(self._h, out.ctypes.data, n))
        return out[:n].tobytes()

    def load(self, blob):
        if len(blob) % 32:
            raise ValueError("index blob must be a multiple of 32 bytes")
        arr = np.frombuffer(blob, dtype=np.uint8)
        n_new = C.c_uint64()
        self._eng._check(self._lib.mi_index_import(self._h, arr.ctypes.data if arr.size else None,
                                                   len(blob) // 32, C.byref(n_new)))
        return n_new.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi_index_free(self._h)
            self._h = None

    free = close
    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def default_config(**overrides):
    cfg = Config()
    rc = load_library().mi_config_default(C.byref(cfg))
    if rc:
        raise MiError(rc, "mi_config_default")
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


class Engine:
    """One mi_ctx: a device, its streams and the Gear parameters."""

    def __init__(self, **cfg_overrides):
        self._lib = load_library()
        self.cfg = default_config(**cfg_overrides)
        h = C.c_void_p()
        rc = self._lib.mi_ctx_create(C.byref(self.cfg), C.byref(h))
        if rc:
            raise MiError(rc, self._lib.mi_last_error(None).decode())
        self._h = h
        self._children = weakref.WeakSet()     # live batches / indexes: they die before the ctx

    def _check(self, rc):
        if rc:
            raise MiError(rc, self._lib.mi_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            for child in list(self._children):  # the library refuses to destroy a ctx with live children
                child.free()
            rc = self._lib.mi_ctx_destroy(self._h)
            if rc:
                raise MiError(rc, self._lib.mi_last_error(self._h).decode())
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def warm(self):
        """mi_ctx_warm: the reader threads, their pinned slabs and the kernels' code objects NOW instead of inside the first commit"""
        self._check(self._lib.mi_ctx_warm(self._h))

    def device_info(self):
        ncu, mhz, mem = C.c_int32(), C.c_int32(), C.c_uint64()
        name = C.create_string_buffer(256)
        self._check(self._lib.mi_device_info(self._h, C.byref(ncu), C.byref(mhz), C.byref(mem),
                                             name, 256))
        return {"n_cu": ncu.value, "clock_mhz": mhz.value, "hbm_bytes": mem.value,
                "name": name.value.decode()}

    def sha_valu_roof(self, waves_per_simd=0, blocks=0):
        """The SHA-256 VALU roof of this device right now, in bytes hashed per second (mi_sha_valu_roof)."""
        v = C.c_double()
        self._check(self._lib.mi_sha_valu_roof(self._h, waves_per_simd, blocks, C.byref(v)))
        return v.value

    def debug_sha_wave_stats(self, path):
        """mi_debug_sha_wave_stats: per-wave records of every chunk pass of THIS ctx go to `path` (None: off)."""
        self._check(self._lib.mi_debug_sha_wave_stats(self._h, os.fsencode(path) if path else None))

    def stats(self):
        st = Stats()
        self._check(self._lib.mi_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def batch(self, n_files_hint=0, bytes_hint=0):
        return Batch(self, n_files_hint, bytes_hint)

    def sha256_many(self, blobs):
        """Batched image.Digester.FromBytes: list of bytes-like -> list of 32-byte digests."""
        lens = np.array([len(b) for b in blobs], dtype=np.uint64)
        offs = np.zeros(len(blobs), dtype=np.uint64)
        if len(blobs):
            offs[1:] = np.cumsum(lens)[:-1]
        data = np.frombuffer(b"".join(bytes(b) for b in blobs), dtype=np.uint8)
        out = np.zeros((len(blobs), 32), dtype=np.uint8)
        u64p = C.POINTER(C.c_uint64)
        self._check(self._lib.mi_sha256_many(self._h, data.ctypes.data if data.size else None,
                                             offs.ctypes.data_as(u64p), lens.ctypes.data_as(u64p),
                                             len(blobs), out.ctypes.data))
        return [out[i].tobytes() for i in range(len(blobs))]

    def index(self, capacity_hint=0):
        """A chunk-digest set that outlives batches (dedup across layers)."""
        return ChunkIndex(self, capacity_hint)

    # ---- native RCCL exchange (what a Go host would use; bench.py drives torch instead) ----
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        rc = load_library().mi_comm_unique_id(buf)
        if rc:
            raise MiError(rc, load_library().mi_last_error(None).decode())
        return buf.raw

    def comm_init_rank(self, nranks, rank, uid):
        self._check(self._lib.mi_comm_init_rank(self._h, nranks, rank, uid))

    def comm_destroy(self):
        self._check(self._lib.mi_comm_destroy(self._h))

    def comm_ranks(self):
        """Ranks of this ctx's communicator as the collective library counts them (0: none)."""
        n = C.c_int()
        self._check(self._lib.mi_comm_ranks(self._h, C.byref(n)))
        return n.value

    def comm_exchange_ms(self):
        """(ms_gather, ms_marking) of this ctx's last exchange, device time from HIP events."""
        a, b = C.c_double(), C.c_double()
        self._check(self._lib.mi_comm_exchange_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def dedup_mark_range(self, d_digests_ptr, n_total, own_first, own_n, d_dup_of_own_ptr):
        """dup_of (global indices) for the rows [own_first, own_first+own_n) of a job-wide,
        rank-major digest set; returns how many of them are job-wide first occurrences."""
        nf = C.c_uint64()
        self._check(self._lib.mi_dedup_mark_range(self._h, d_digests_ptr, n_total, own_first, own_n,
                                                  d_dup_of_own_ptr, C.byref(nf)))
        return nf.value

    def dedup_mark(self, d_digests_ptr, n, d_dup_of_ptr):
        """dup_of over a device-resident digest set (e.g. the all-gathered one)."""
        nu = C.c_uint64()
        self._check(self._lib.mi_dedup_mark(self._h, d_digests_ptr, n, d_dup_of_ptr, C.byref(nu)))
        return nu.value


def comm_init_all(engines):
    """mi_comm_init_all: one communicator over the ctxs of ONE process (rank i = engines[i])."""
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    rc = load_library().mi_comm_init_all(arr, len(engines))
    if rc:
        raise MiError(rc, engines[0]._lib.mi_last_error(engines[0]._h).decode())


def dedup_allgather_all(batches, form="allgather"):
    """mi_dedup_allgather_all (form="alltoall": mi_dedup_alltoall_all, the hash-partitioned form -- same results): the
    exchange + job-wide marking for the batches of comm_init_all's ctxs, rank order.  Returns (n_total, n_unique)."""
    arr = (C.c_void_p * len(batches))(*[b._h for b in batches])
    a, b_ = C.c_uint64(), C.c_uint64()
    fn = {"allgather": "mi_dedup_allgather_all", "alltoall": "mi_dedup_alltoall_all"}[form]
    rc = getattr(load_library(), fn)(arr, len(batches), C.byref(a), C.byref(b_))
    if rc:
        e = batches[0].engine
        raise MiError(rc, "; ".join(x.engine._lib.mi_last_error(x.engine._h).decode() for x in batches) or str(e))
    return a.value, b_.value


class Batch:
    def __init__(self, engine, n_files_hint=0, bytes_hint=0):
        self.engine = engine
        self._lib = engine._lib
        h = C.c_void_p()
        engine._check(self._lib.mi_batch_begin(engine._h, n_files_hint, bytes_hint, C.byref(h)))
        self._h = h
        engine._children.add(self)

    def _check(self, rc):
        self.engine._check(rc)

    def add_bytes(self, data, tag=0):
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else \
            np.ascontiguousarray(data).view(np.uint8)
        self._check(self._lib.mi_batch_add_bytes(self._h, a.ctypes.data if a.size else None,
                                                 a.size, tag))

    def reserve(self, more_files, more_bytes):
        """mi_batch_reserve: room for what is known to come, so that the arena does not grow under way."""
        self._check(self._lib.mi_batch_reserve(self._h, more_files, more_bytes))

    def add_path(self, path, size=None, tag=0):
        if size is None:
            size = os.stat(path).st_size
        self._check(self._lib.mi_batch_add_path(self._h, os.fsencode(path), size, tag))

    def add_paths(self, paths, sizes=None, tags=None):
        """Many files at once, opened by the reader threads (deferred: a missing file fails run())."""
        n = len(paths)
        if sizes is None:
            sizes = [os.stat(p).st_size for p in paths]
        arr = (C.c_char_p * max(n, 1))(*[os.fsencode(p) for p in paths])
        sz = np.ascontiguousarray(sizes, dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        tg = np.ascontiguousarray(tags, dtype=np.uint64).ctypes.data_as(u64p) if tags is not None else None
        self._check(self._lib.mi_batch_add_paths(self._h, n, arr, sz.ctypes.data_as(u64p), tg))

    def add_path_range(self, path, offset, size, tag=0):
        """A file that is bytes [offset, offset+size) of `path` (a member of a layer tar)."""
        self._check(self._lib.mi_batch_add_path_range(self._h, os.fsencode(path), offset, size, tag))

    def add_tar(self, path):
        """Registers every regular file of an uncompressed layer tar straight from its byte range
        in the archive (no extraction).  Returns the archive's entries (tar_entries) with
        "file_index" rewritten to the files' indices in THIS batch."""
        ents = tar_entries(path)
        base = self.counts()[0]
        for e in ents:
            if e["kind"] == KIND_FILE:
                self.add_path_range(path, e["data_offset"], e["size"], tag=e["file_index"])
                e["file_index"] += base
        return ents

    def add_synthetic(self, sizes, content_ids=None, seed=0x4D414B49):
        s = np.ascontiguousarray(sizes, dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        cp = None
        if content_ids is not None:
            cids = np.ascontiguousarray(content_ids, dtype=np.uint64)
            assert cids.size == s.size
            cp = cids.ctypes.data_as(u64p)
        self._check(self._lib.mi_batch_add_synthetic(self._h, s.size, s.ctypes.data_as(u64p), cp,
                                                     seed))

    # ---- parts: one file split across batches / GPUs ------------------------------------
    def add_path_part(self, path, begin, end, file_size=None, tag=0):
        if file_size is None:
            file_size = os.stat(path).st_size
        self._check(self._lib.mi_batch_add_path_part(self._h, os.fsencode(path), file_size, begin, end, tag))

    def add_synthetic_part(self, file_size, content_id, begin, end, seed=0x4D414B49):
        self._check(self._lib.mi_batch_add_synthetic_part(self._h, file_size, content_id, seed, begin, end))

    def scan_cuts(self):
        """Blocking: stage + Gear marking + cut selection only (the parts' exits become known)."""
        self._check(self._lib.mi_batch_scan_cuts(self._h))
        return self

    def parts(self):
        n = C.c_uint64()
        self._check(self._lib.mi_batch_parts(self._h, None, 0, C.byref(n)))
        arr = (PartState * max(n.value, 1))()
        self._check(self._lib.mi_batch_parts(self._h, arr, n.value, None))
        return [{k: getattr(arr[i], k) for k, _ in PartState._fields_} for i in range(n.value)]

    def set_part_entry(self, file_index, entry):
        self._check(self._lib.mi_batch_set_part_entry(self._h, file_index, entry))

    def fix_cuts(self):
        self._check(self._lib.mi_batch_fix_cuts(self._h))
        return self

    def run(self):
        self._check(self._lib.mi_batch_run(self._h))
        return self

    def rerun(self):
        self._check(self._lib.mi_batch_rerun(self._h))
        return self

    def submit(self):
        """Enqueue the pipeline on the batch's own stream and return (no host sync)."""
        self._check(self._lib.mi_batch_submit(self._h))
        return self

    def wait(self):
        self._check(self._lib.mi_batch_wait(self._h))
        return self

    def counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.mi_batch_counts(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def files(self):
        n = self.counts()[0]
        out = np.zeros(max(n, 1), dtype=FILE_DTYPE)
        self._check(self._lib.mi_batch_files(self._h, out.ctypes.data, n))
        return out[:n]

    def chunks(self):
        n = self.counts()[1]
        out = np.zeros(max(n, 1), dtype=CHUNK_DTYPE)
        self._check(self._lib.mi_batch_chunks(self._h, out.ctypes.data, n))
        return out[:n]

    def files_view(self):
        """The file rows WITHOUT a copy (same lifetime as chunks_view's)."""
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.mi_batch_files_view(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=FILE_DTYPE)
        buf = (C.c_char * (n.value * FILE_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=FILE_DTYPE)

    def roots(self):
        """the per-file chunk roots alone: an (n_files, 32) uint8 array"""
        n = self.counts()[0]
        out = np.zeros((max(n, 1), 32), dtype=np.uint8)
        self._check(self._lib.mi_batch_roots(self._h, out.ctypes.data, n))
        return out[:n]

    def read_file(self, file_index, offset=0, length=None):
        """bytes [offset, offset + length) of a staged file as they lie in HBM"""
        if length is None:
            raise ValueError("length")
        out = np.zeros(max(length, 1), dtype=np.uint8)
        self._check(self._lib.mi_batch_read_file(self._h, file_index, offset, out.ctypes.data, length))
        return out[:length].tobytes()

    def chunks_view(self):
        """The chunk rows WITHOUT a copy: a numpy array over the batch's own pinned buffer, valid
        until the batch is submitted again, marked globally, reset or freed."""
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.mi_batch_chunks_view(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=CHUNK_DTYPE)
        buf = (C.c_char * (n.value * CHUNK_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=CHUNK_DTYPE)

    def device_digests(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.mi_batch_device_digests(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def add_tree(self, root, rel_base=None, blacklist=(), mode=TREE_CONTEXT):
        """Walk `root` the way the reference does (filepath.Walk order) and add its regular
        files; returns the number of recorded entries so far."""
        bl = (C.c_char_p * max(len(blacklist), 1))(*[os.fsencode(x) for x in blacklist])
        n = C.c_uint64()
        self._check(self._lib.mi_batch_add_tree(self._h, os.fsencode(root),
                                                os.fsencode(rel_base) if rel_base else None,
                                                bl, len(blacklist), mode, C.byref(n)))
        return n.value

    def tree_entries(self, n):
        arr = (TreeEntry * max(n, 1))()
        self._check(self._lib.mi_batch_tree_entries(self._h, arr, n))
        return [(os.fsdecode(e.relpath), os.fsdecode(e.link_target) if e.link_target else None,
                 e.file_index, e.size, e.kind, e.mode) for e in arr[:n]]

    def context_checksum_tree(self, prefix):
        pre = bytes(prefix)
        out = C.c_uint32()
        self._check(self._lib.mi_context_checksum_tree(self._h, pre, len(pre), C.byref(out)))
        return "%x" % out.value

    def context_checksum(self, prefix, entries):
        """The reference's COPY/ADD running CRC32 (add_copy_step.go:102-238).

        entries: walk-ordered list of (relpath, link_target_or_None, file_index_or_-1);
        returns the cache ID the way SetCacheID prints it ("%x", unpadded)."""
        arr = (CtxEntry * max(len(entries), 1))()
        for i, (rel, link, idx) in enumerate(entries):
            arr[i].relpath = os.fsencode(rel)
            arr[i].link_target = os.fsencode(link) if link is not None else None
            arr[i].file_index = idx
        pre = bytes(prefix)
        out = C.c_uint32()
        self._check(self._lib.mi_context_checksum(self._h, pre, len(pre), arr, len(entries),
                                                  C.byref(out)))
        return "%x" % out.value

    def dedup_allgather(self):
        """Collective: RCCL all-gather of the digest sets + global marking inside the library.
        Returns (n_total, n_unique, first_global)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.mi_dedup_allgather(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def dedup_alltoall(self):
        """Collective: the hash-partitioned form of dedup_allgather (mi_dedup_alltoall) -- every digest travels to ONE
        owner rank and an 8-byte answer comes back.  Same results: (n_total, n_unique, first_global)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.mi_dedup_alltoall(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def mark_global(self, d_digests_all_ptr, n_total, own_first):
        """Job-wide marking of this batch's chunks (rows own_first.. of the gathered set) straight
        into its dup_of column; returns how many of them are job-wide first occurrences."""
        nf = C.c_uint64()
        self._check(self._lib.mi_batch_mark_global(self._h, d_digests_all_ptr, n_total, own_first,
                                                   C.byref(nf)))
        return nf.value

    def device_dup_of(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.mi_batch_device_dup_of(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_global_dedup(self, d_dup_of_global_ptr, first_global):
        self._check(self._lib.mi_batch_set_global_dedup(self._h, d_dup_of_global_ptr, first_global))

    def read_back(self):
        n = self.counts()[2]
        out = np.zeros(max(n, 1), dtype=np.uint8)
        self._check(self._lib.mi_batch_read_back(self._h, out.ctypes.data, n))
        return out[:n]

    def stage_stats(self):
        """Staging counters (FLAG_VERIFY_STAGING) + "note": what the first mismatch looked like."""
        st = StageStats()
        self._check(self._lib.mi_batch_stage_stats(self._h, C.byref(st)))
        d = st.as_dict()
        d["note"] = self._lib.mi_batch_stage_note(self._h).decode(errors="replace")
        return d

    def reset(self):
        """Empty the batch, keep its device memory (next layer, same buffers)."""
        self._check(self._lib.mi_batch_reset(self._h))
        return self

    def free(self):
        if getattr(self, "_h", None):
            self._lib.mi_batch_free(self._h)
            self._h = None

    __del__ = free

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()
