#!/usr/bin/env python3
"""Writes tests/golden/gear_cdc_kat.json: known answers of THIS repo's Gear-CDC spec, produced
by the CPU oracle.  They pin the spec against drift; they are NOT reference outputs (the
reference has no CDC -- parity unpinned)."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import mi_oracle as O  # noqa: E402

CASES = [("c2_file0", 0x4D414B49, 13, 2048, 65536, 0x4D414B49, 0, 65536),
         ("c2_file1", 0x4D414B49, 13, 2048, 65536, 0x4D414B49, 1, 65536),
         ("one_mib", 0x4D414B49, 13, 2048, 65536, 0x4D414B49, 7, 1 << 20),
         ("small_chunks", 0x4D414B49, 8, 64, 1024, 0x4D414B49, 2, 20000),
         ("other_seed", 12345, 10, 1024, 8192, 99, 3, 100000),
         ("forced_only", 0x4D414B49, 32, 2048, 4096, 0x4D414B49, 4, 20000),
         ("tiny", 0x4D414B49, 13, 2048, 65536, 0x4D414B49, 5, 100)]


def main():
    out = []
    for name, gs, mb, mn, mx, ds, cid, size in CASES:
        p = O.CdcParams(gs, mb, mn, mx)
        data = O.synth_fill(ds, cid, 0, size)
        ends = O.cdc_two_phase(data, p)
        files, _ = O.scan_batch(data, [0], [size], p)
        out.append({"name": name, "gear_seed": gs, "mask_bits": mb, "min_size": mn, "max_size": mx,
                    "data_seed": ds, "content_id": cid, "size": size,
                    "ends": [int(x) for x in ends],
                    "chunk_root": files["chunk_root"][0].tobytes().hex(),
                    "sha256": hashlib.sha256(data.tobytes()).hexdigest()})
    json.dump({"comment": "Gear-CDC known answers of this repo's own spec (oracle-generated)",
               "vectors": out}, open(os.path.join(HERE, "gear_cdc_kat.json"), "w"), indent=1)
    print("wrote", len(out), "vectors")


if __name__ == "__main__":
    main()
