"""Debug: which chunks of a C4 shard does the engine call duplicates, and are their bytes equal?"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd  # noqa: E402
from makisu_amd import workloads as W  # noqa: E402
from oracle import mi_oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
sh = W.c4(0, 1, n)
with makisu_amd.Engine() as e, e.batch(sh.n_files, sh.n_bytes) as b:
    b.add_synthetic(sh.sizes, sh.cids, seed=sh.seed)
    b.run()
    chunks = b.chunks().copy()
    files = b.files().copy()
dups = np.nonzero(chunks["dup_of"] >= 0)[0]
print("files", n, "chunks", len(chunks), "dups", len(dups), "sum n_chunks", int(files["n_chunks"].sum()))
for i in dups[:30]:
    a, t = chunks[i], chunks[chunks["dup_of"][i]]
    ba = O.synth_fill(sh.seed, int(sh.cids[a["file_index"]]), int(a["offset"]), int(a["length"])).tobytes()
    bt = O.synth_fill(sh.seed, int(sh.cids[t["file_index"]]), int(t["offset"]), int(t["length"])).tobytes()
    print("row %d (file %d off %d len %d) -> row %d (file %d off %d len %d): bytes equal %s, gpu digests equal %s, sha ok %s %s"
          % (i, a["file_index"], a["offset"], a["length"], chunks["dup_of"][i], t["file_index"], t["offset"], t["length"],
             ba == bt, a["sha256"].tobytes() == t["sha256"].tobytes(),
             hashlib.sha256(ba).digest() == a["sha256"].tobytes(), hashlib.sha256(bt).digest() == t["sha256"].tobytes()))
# tiling check
ends = chunks["offset"] + chunks["length"]
same = chunks["file_index"][1:] == chunks["file_index"][:-1]
print("contiguous within files:", bool((chunks["offset"][1:][same] == ends[:-1][same]).all()),
      "file order monotone:", bool((np.diff(chunks["file_index"].astype(np.int64)) >= 0).all()))
