"""The host rows at C2's and C4's entry counts (a3 / a4 / a7 / a8 / a13 in ENTRIES -- 100 000 for C2, ten million
for C4): entry arrays are built with numpy and handed straight to the C ABI, so what is timed is the library, not the
python harness.  No GPU, no files on disk (the entries are made up; a scan's deletion check never reaches lstat because
every path is listed).  usage: host_scale_bench.py [entries = 1000000]      -> profiles/<tag>_host_scale.txt"""
import ctypes as C
import os
import resource
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import makisu_amd as M  # noqa: E402

T = M.TreeEntry
DT = np.dtype({"names": ["relpath", "link", "file_index", "size", "mtime", "mode", "kind", "uid", "gid"],
               "formats": ["<u8", "<u8", "<i8", "<u8", "<i8", "<u4", "u1", "<u4", "<u4"],
               "offsets": [T.relpath.offset, T.link_target.offset, T.file_index.offset, T.size.offset,
                           T.mtime_sec.offset, T.mode.offset, T.kind.offset, T.uid.offset, T.gid.offset],
               "itemsize": C.sizeof(T)})


def entries(names):
    """names (bytes, directories = no '/') -> (numpy array laid out as mi_tree_entry[], keep-alive)"""
    n = len(names)
    buf = b"\0".join(names) + b"\0"
    lens = np.fromiter((len(x) + 1 for x in names), dtype=np.int64, count=n)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    raw = C.create_string_buffer(buf, len(buf))
    arr = np.zeros(n, dtype=DT)
    arr["relpath"] = C.addressof(raw) + offs
    isdir = np.fromiter((b"/" not in x for x in names), dtype=bool, count=n)
    arr["kind"] = np.where(isdir, 0, 1)
    arr["mode"] = np.where(isdir, 0o40755, 0o100644)
    arr["size"] = np.where(isdir, 0, 65536)
    arr["mtime"] = 1600000000
    arr["file_index"] = -1
    return arr, raw


def ptr(arr):
    return C.cast(arr.ctypes.data, C.POINTER(T))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    L = M.load_library()
    ndirs = max(1, n // 100)
    names = [b"d%05d" % i for i in range(ndirs)] + [b"d%05d/f%08d.bin" % (i % ndirs, i) for i in range(n)]
    names.sort(key=lambda p: p.split(b"/"))                   # filepath.Walk order
    N = len(names)
    print("%d entries (%d directories of 100 files), walk order; %s" % (N, ndirs, time.strftime("%Y-%m-%d")))

    a, keep_a = entries(names)
    out = np.zeros(N, dtype=np.uint64)
    t0 = time.perf_counter()
    assert L.mi_entries_commit_order(ptr(a), N, C.cast(out.ctypes.data, C.POINTER(C.c_uint64))) == 0
    print("mi_entries_commit_order, walk-ordered input   %8.3f s  %6.3f us / entry" % (time.perf_counter() - t0, (time.perf_counter() - t0) / N * 1e6))
    rng = np.random.default_rng(1)
    perm = rng.permutation(N)
    sh, keep_s = entries([names[i] for i in perm])
    t0 = time.perf_counter()
    assert L.mi_entries_commit_order(ptr(sh), N, C.cast(out.ctypes.data, C.POINTER(C.c_uint64))) == 0
    print("mi_entries_commit_order, shuffled input       %8.3f s  %6.3f us / entry" % (time.perf_counter() - t0, (time.perf_counter() - t0) / N * 1e6))
    del sh, keep_s

    b, keep_b = entries(names)
    b["mtime"][::1000] += 5                                    # one entry in a thousand changed
    root = tempfile.mkdtemp()
    with M.MemFS(root) as fs:
        nm = C.c_uint64()
        t0 = time.perf_counter()
        assert L.mi_memfs_update_from_entries(fs._h, ptr(a), N, C.byref(nm)) == 0
        dt = time.perf_counter() - t0
        print("mi_memfs_update_from_entries (layer merge)    %8.3f s  %6.3f us / entry   (%d merged)" % (dt, dt / N * 1e6, nm.value))
        lay, ne = C.c_void_p(), C.c_uint64()
        t0 = time.perf_counter()
        assert L.mi_memfs_add_layer_by_scan(fs._h, ptr(b), N, None, 0, C.byref(lay), C.byref(ne)) == 0
        dt = time.perf_counter() - t0
        print("mi_memfs_add_layer_by_scan, 0.1 %% changed      %8.3f s  %6.3f us / entry   (layer of %d)" % (dt, dt / N * 1e6, ne.value))
        L.mi_copy_layer_free(lay)
        t0 = time.perf_counter()
        assert L.mi_memfs_add_layer_by_scan(fs._h, ptr(b), N, None, 0, C.byref(lay), C.byref(ne)) == 0
        dt = time.perf_counter() - t0
        print("mi_memfs_add_layer_by_scan, nothing changed   %8.3f s  %6.3f us / entry   (layer of %d)" % (dt, dt / N * 1e6, ne.value))
        L.mi_copy_layer_free(lay)
    os.rmdir(root)

    sa = M.SnapshotSide(ptr(a), N, None, 0, None)
    sb = M.SnapshotSide(ptr(b), N, None, 0, None)
    fl = np.zeros(N, dtype=np.uint8)
    wh = np.zeros(N, dtype=np.uint8)
    t0 = time.perf_counter()
    assert L.mi_snapshot_diff(C.byref(sa), C.byref(sb), 0, C.cast(fl.ctypes.data, C.POINTER(C.c_uint8)), C.cast(wh.ctypes.data, C.POINTER(C.c_uint8))) == 0
    dt = time.perf_counter() - t0
    print("mi_snapshot_diff (stateless twin)             %8.3f s  %6.3f us / entry   (%d changed, %d carried)" % (dt, dt / N * 1e6, int((fl == 1).sum()), int((fl == 2).sum())))

    m = min(N, 200000)
    d, keep_d = entries([b"usr/lib/d%05d" % i for i in range(m)])      # header-only members (directories)
    with M.Layer(out_fd=-1, gzip_level=M.GZIP_OFF) as layer:
        add = L.mi_layer_add
        view = (T * m).from_buffer(d)
        t0 = time.perf_counter()
        for i in range(m):
            add(layer._h, C.byref(view[i]), None)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        for i in range(m):
            C.byref(view[i])
            L.mi_abi_version()
        call = (time.perf_counter() - t1) / m
    print("mi_layer_add, header-only members             %8.3f s  %6.3f us / entry   (%d members; the python call itself %.2f us)" % (dt, dt / m * 1e6, m, call * 1e6))
    print("peak resident set %.1f GB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))


if __name__ == "__main__":
    main()
