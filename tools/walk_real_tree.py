"""The snapshot walk on a real root file system: this image's / with every top-level directory but /usr blacklisted (the skip rules asked of every path),
and /usr walked without skip rules (MI_TREE_CONTEXT), three and two times.   usage: walk_real_tree.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd as M
bl = ["/" + n for n in os.listdir("/") if n != "usr"]
for k in range(3):
    t0 = time.perf_counter()
    L = M.load_library()
    import ctypes as C
    arr = (C.c_char_p * len(bl))(*[os.fsencode(x) for x in bl])
    h, n = C.c_void_p(), C.c_uint64()
    rc = L.mi_tree_walk(b"/", b"/", arr, len(bl), M.TREE_SCAN, C.byref(h), C.byref(n))
    dt = time.perf_counter() - t0
    print("walk of / with %d blacklisted top-level entries: rc %d, %d entries in %.3f s (%.2f us each)" % (len(bl), rc, n.value, dt, dt * 1e6 / max(1, n.value)))
    L.mi_tree_free(h)
for k in range(2):
    t0 = time.perf_counter()
    h, n = C.c_void_p(), C.c_uint64()
    rc = L.mi_tree_walk(b"/usr", b"/usr", None, 0, M.TREE_CONTEXT, C.byref(h), C.byref(n))
    dt = time.perf_counter() - t0
    print("context walk of /usr (no skip rules but special files): rc %d, %d entries in %.3f s (%.2f us each)" % (rc, n.value, dt, dt * 1e6 / max(1, n.value)))
    L.mi_tree_free(h)
