"""BASELINE.json configs[4] shape on one GPU: Zipf-ish sizes 1 KiB..1 GiB, 90 % duplicate files."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd  # noqa: E402


def main():
    total_target = int(sys.argv[1]) if len(sys.argv) > 1 else 32 << 30
    rng = np.random.default_rng(0x4D414B49 + 2)
    sizes = []
    while sum(sizes) < total_target // 10:                  # the 10 % distinct contents
        sizes.append(int(2.0 ** rng.uniform(10, 30)))
    distinct = len(sizes)
    sizes = np.array(sizes, dtype=np.int64)
    src = rng.integers(0, distinct, 9 * distinct)
    all_sizes = np.concatenate([sizes, sizes[src]])
    cids = np.concatenate([np.arange(distinct), src])
    keep = np.cumsum(all_sizes) <= total_target
    all_sizes, cids = all_sizes[keep], cids[keep]
    with makisu_amd.Engine() as e, e.batch() as b:
        b.add_synthetic(all_sizes.astype(np.uint64), cids.astype(np.uint64), seed=0x4D414B49 + 2)
        b.run()
        for _ in range(2):
            t0 = time.perf_counter()
            b.rerun()
            dt = time.perf_counter() - t0
        st = e.stats()
        print(json.dumps({"files": int(len(all_sizes)), "distinct_contents": int(distinct),
                          "bytes": int(all_sizes.sum()), "max_file": int(all_sizes.max()),
                          "ms_per_pass": round(dt * 1e3, 2), "GiBps": round(all_sizes.sum() / dt / 2**30, 1),
                          "chunks": st["n_chunks"], "unique_chunks": st["n_unique"],
                          "phases_ms": {k: round(v, 2) for k, v in st.items() if k.startswith("ms_")}}))


if __name__ == "__main__":
    main()
