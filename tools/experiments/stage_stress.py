"""Stress for an intermittent mismatch seen once in test_host_fed_mix_large_and_small_files."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402
import makisu_amd  # noqa: E402
from oracle import mi_oracle as O  # noqa: E402

SEED = 0x4D414B49
big = 1 << 30
sizes = [big, 700, big, 65536, 0, big, 1, 4097, big] + [int(x) for x in np.random.default_rng(4).integers(1, 200000, 300)]
cids = list(range(6000, 6000 + len(sizes)))
data, offs = O.synth_fill_many(SEED + 1, cids, sizes, 8)
d = tempfile.mkdtemp(dir="/dev/shm")
paths = []
for i, (o, n) in enumerate(zip(offs, sizes)):
    p = os.path.join(d, "f%04d" % i)
    data[int(o):int(o) + n].tofile(p)
    paths.append(p)
extra = [O.synth_fill(SEED, 6500, 0, 5 << 20).tobytes(), b"tiny", O.synth_fill(SEED, 6501, 0, 90000).tobytes()]
ref = None
bad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
with makisu_amd.Engine() as e:
    for rep in range(reps):
        with e.batch() as b:
            for i, (pth, n) in enumerate(zip(paths, sizes)):
                b.add_path(pth, n, i)
                if i == 5:
                    for x in extra:
                        b.add_bytes(x)
            b.run()
            fl = b.files().copy()
            back = None
            if ref is None:
                ref = fl
            elif not np.array_equal(fl["chunk_root"], ref["chunk_root"]):
                bad += 1
                diff = np.nonzero((fl["chunk_root"] != ref["chunk_root"]).any(axis=1))[0]
                back = b.read_back()
                starts = np.concatenate([[0], np.cumsum(fl["size"])[:-1]])
                for f in diff[:4]:
                    s0, n = int(starts[f]), int(fl["size"][f])
                    got = back[s0:s0 + n]
                    if f < 6:
                        want = data[int(offs[f]):int(offs[f]) + n]
                    elif f < 9:
                        want = np.frombuffer(extra[f - 6], dtype=np.uint8)
                    else:
                        want = data[int(offs[f - 3]):int(offs[f - 3]) + n]
                    neq = np.nonzero(got != want)[0]
                    print("rep", rep, "file", int(f), "size", n, "bytes differ:", len(neq),
                          "first", int(neq[0]) if len(neq) else None, "last", int(neq[-1]) if len(neq) else None,
                          "got zeros there:", bool(len(neq) and not got[neq].any()), flush=True)
    print("reps", reps, "mismatching runs", bad)
for p in paths:
    os.unlink(p)
os.rmdir(d)
