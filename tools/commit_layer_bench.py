"""step.commitLayer on the host rows alone (no GPU): a tree of n files of `bytes` bytes in /dev/shm, one MemFS handle,
three commits by scan -- everything new (walk + scan + every file framed into the layer tar and digested), nothing
changed (walk + scan: the empty layer), one file in ten directories touched.  MI_WALK_TIMING / MI_MEMFS_TIMING lines show
where the time of each went.  usage: commit_layer_bench.py [files = 100000] [bytes = 4096]"""
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd as M  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    root = "/dev/shm/mi_commit_layer_%d_%d_%d" % (n, size, os.getpid())
    os.makedirs(root)
    try:
        blob = os.urandom(size)
        for d in range(max(1, n // 200)):
            dn = os.path.join(root, "d%04d" % d)
            os.mkdir(dn)
            for k in range(200):
                with open(os.path.join(dn, "f%03d" % k), "wb") as f:
                    f.write(blob)
        os.environ.setdefault("MI_WALK_TIMING", "1")
        os.environ.setdefault("MI_MEMFS_TIMING", "1")
        with M.MemFS(root) as fs:
            for step, what in enumerate(("everything new", "nothing changed", "one file in every tenth directory appended to")):
                if step == 2:
                    for d in range(0, max(1, n // 200), 10):
                        with open(os.path.join(root, "d%04d" % d, "f000"), "ab") as f:
                            f.write(b"x")
                t0 = time.perf_counter()
                res = fs.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF)
                dt = time.perf_counter() - t0
                print("commit %d (%s): %.3f s -> %d entries, %d tar bytes, TarDigest %s" %
                      (step, what, dt, res["n_entries"], res["tar_bytes"], str(res["tar_digest"])[:19]), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
