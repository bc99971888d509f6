"""Randomized differential soak on the GPU: batches of random shape, content and CDC parameters through the C ABI against
the oracle, every column (cut points, chunk digests, file roots, duplicate marking, the staged bytes), for a time budget.
What the fixed cases of tests/test_gpu_parity.py cannot enumerate: sizes around every tile / group / block boundary at
once, low-entropy runs inside random data, parameter sets nobody wrote down.  A failing case is printed with the seed
that reproduces it.  No torch.
    python tools/gpu_fuzz.py [seconds: 60] [seed: 1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd  # noqa: E402
from oracle import mi_oracle as oracle  # noqa: E402

EDGES = [0, 1, 55, 56, 63, 64, 65, 119, 120, 127, 128, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 16383, 16384,
         65535, 65536, 65537, 65536 + 63, 65536 + 64, 131071, 131072, 131073, 262143, 262144, 262145, 262144 + 65536,
         524288, 1048576 + 1]


def content(rng, n, ident):
    kind = rng.integers(0, 10)
    if n == 0:
        return b""
    if kind <= 4:
        return oracle.synth_fill(0x4D414B49, int(ident), 0, int(n)).tobytes()
    if kind == 5:
        return bytes(int(n))                                   # all zero: forced cuts only
    if kind == 6:
        p = rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8).tobytes()
        return (p * (int(n) // len(p) + 1))[:int(n)]            # periodic
    if kind == 7:                                              # random with a zero run somewhere inside
        a = bytearray(rng.integers(0, 256, int(n), dtype=np.uint8).tobytes())
        lo = int(rng.integers(0, n))
        hi = min(int(n), lo + int(rng.integers(1, 200000 if n < (1 << 20) else 3000000)))
        a[lo:hi] = bytes(hi - lo)
        return bytes(a)
    if kind == 8:
        return bytes([int(rng.integers(0, 256))]) * int(n)
    return rng.integers(0, 4, int(n), dtype=np.uint8).tobytes()  # two bits of entropy per byte


def one_case(rng, case):
    mask_bits = int(rng.choice([0, 1, 4, 6, 8, 10, 11, 12, 13, 14, 16, 20, 32]))
    min_size = int(rng.choice([64, 65, 100, 256, 1000, 2048, 4096, 10000]))
    max_size = int(max(min_size, rng.choice([64, 256, 1024, 4096, 8192, 65536, 65537, 100000, 262144, 300000, 1 << 20])))
    n_files = int(rng.choice([1, 2, 7, 40, 300]))
    cap = int(rng.choice([3000, 70000, 300000, 2500000]))
    sizes = []
    for _ in range(n_files):
        r = rng.integers(0, 4)
        if r == 0:
            sizes.append(int(rng.choice(EDGES)) + int(rng.integers(-1, 2)) * int(rng.integers(0, 2)))
        elif r == 1:
            sizes.append(int(rng.integers(0, 300)))
        else:
            sizes.append(int(rng.integers(0, cap)))
    if rng.integers(0, 6) == 0:                                 # a few LARGE files: groups, speculation, the per-file fix-up
        sizes = [int(rng.integers(1 << 20, 36 << 20)) + int(rng.choice([0, 1, -1, 65536, 262144])) * int(rng.integers(0, 2))
                 for _ in range(int(rng.integers(1, 4)))]
    sizes = [max(0, s) for s in sizes]
    while sum(sizes) > 40_000_000:
        sizes[int(np.argmax(sizes))] //= 2
    blobs = [content(rng, s, case * 1000 + i) for i, s in enumerate(sizes)]
    if len(blobs) > 2 and rng.integers(0, 2):                      # duplicates: whole files and prefixes
        blobs[-1] = blobs[0]
        blobs[-2] = blobs[0][:len(blobs[0]) // 2]
    desc = "mask %d min %d max %d, %d files, %d bytes" % (mask_bits, min_size, max_size, len(blobs), sum(len(b) for b in blobs))
    with makisu_amd.Engine(mask_bits=mask_bits, min_size=min_size, max_size=max_size) as e:
        with e.batch(len(blobs), sum(len(b) for b in blobs)) as b:
            for i, blob in enumerate(blobs):
                b.add_bytes(blob, tag=i)
            b.run()
            files, chunks = b.files().copy(), b.chunks().copy()
            back = b.read_back().copy()
        data = np.frombuffer(b"".join(blobs), dtype=np.uint8)
        sz = np.array([len(x) for x in blobs], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(sz)[:-1]]).astype(np.uint64)
        c = e.cfg
        rf, rc = oracle.scan_batch(data if data.size else np.zeros(1, np.uint8), offs, sz,
                                   oracle.CdcParams(c.gear_seed, c.mask_bits, c.min_size, c.max_size))
    bad = []
    if not np.array_equal(back, data):
        bad.append("staged bytes")
    if len(chunks) != len(rc):
        bad.append("chunk count %d vs %d" % (len(chunks), len(rc)))
    else:
        for col in ("file_index", "offset", "length", "sha256", "dup_of"):
            if not np.array_equal(chunks[col], rc[col]):
                bad.append("chunks." + col)
    for col in ("n_chunks", "first_chunk", "chunk_root"):
        if not np.array_equal(files[col], rf[col]):
            bad.append("files." + col)
    return desc, bad, len(rc)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle.build()
    t0, case, total_chunks, total_bytes, failures = time.time(), 0, 0, 0, 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, case])
        desc, bad, n_chunks = one_case(rng, case)
        total_chunks += n_chunks
        if bad:
            failures += 1
            print("FAIL seed %d case %d (%s): %s" % (seed, case, desc, ", ".join(bad)), flush=True)
        case += 1
    print("gpu_fuzz: %d cases, %d chunks compared, %d failure(s) in %.0f s (seed %d)" % (case, total_chunks, failures, time.time() - t0, seed))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
