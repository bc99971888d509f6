"""GPU parity for PARTS: one file split across batches / GPUs (include/makisu_mi.h "parts",
SURVEY.md 8e "files >= 256 MiB split across GPUs").  A part is staged behind a halo, cut under an
assumed entry, and corrected with the previous part's last cut; the parts' chunk rows, put end to
end, must be EXACTLY the rows of the whole file -- checked against the oracle's sequential chunker
and against the engine's own whole-file run.  Cut points: parity UNPINNED w.r.t. the reference (no
CDC there); the oracle is this repo's spec.
"""
import os

import numpy as np
import pytest

try:
    import torch  # noqa: F401  (must come before the engine: see test_gpu_parity.py)
except ImportError:
    torch = None

pytestmark = pytest.mark.gpu

SEED = 0x4D414B49
G = 256 * 1024


def _whole(oracle, eng, data):
    c = eng.cfg
    p = oracle.CdcParams(c.gear_seed, c.mask_bits, c.min_size, c.max_size)
    a = np.frombuffer(data, dtype=np.uint8)
    _, rc = oracle.scan_batch(a, np.array([0], dtype=np.uint64), np.array([len(data)], dtype=np.uint64),
                              p, True, 8, 0)
    return rc


def _check_parts(oracle, eng, data, bounds, add, one_batch=False):
    """add(batch, begin, end) registers the part.  Every part gets its own batch (a GPU each) unless
    one_batch.  Returns the number of exchange rounds."""
    from makisu_amd.distributed import resolve_parts_local
    ref = _whole(oracle, eng, data)
    batches, owners = [], []
    try:
        if one_batch:
            b = eng.batch()
            batches.append(b)
            for k, (lo, hi) in enumerate(bounds):
                add(b, lo, hi)
            owners.append((b, [(0, k) for k in range(len(bounds))]))
        else:
            for k, (lo, hi) in enumerate(bounds):
                b = eng.batch()
                batches.append(b)
                add(b, lo, hi)
                owners.append((b, [(0, k)]))
        rounds = resolve_parts_local(owners)
        rows = []
        for b in batches:
            b.run()
            ch, fl = b.chunks().copy(), b.files().copy()
            for st in b.parts():
                mine = ch[ch["file_index"] == st["file_index"]]
                assert fl["size"][st["file_index"]] == st["end"] - st["begin"]
                assert fl["n_chunks"][st["file_index"]] == len(mine)
                if len(mine):                              # owned: chunks that END in (begin, end]
                    ends = mine["offset"] + mine["length"]
                    assert ends[0] > st["begin"] and ends[-1] <= st["end"]
                    assert mine["offset"][0] == st["entry"] and ends[-1] == st["exit"]
                rows.append(mine)
        got = np.concatenate(rows)
        assert len(got) == len(ref), (len(got), len(ref))
        assert np.array_equal(got["offset"], ref["offset"]), "cut points differ"
        assert np.array_equal(got["length"], ref["length"]), "cut points differ"
        assert np.array_equal(got["sha256"], ref["sha256"]), "chunk digests differ"
        return rounds
    finally:
        for b in batches:
            b.free()


def test_synthetic_parts_equal_the_whole_file(oracle):
    import makisu_amd
    from makisu_amd.workloads import split_file
    n = 21 * G + 12345
    data = oracle.synth_fill(SEED, 4100, 0, n).tobytes()
    with makisu_amd.Engine() as e:
        for n_parts in (2, 3, 8):
            bounds = split_file(n, n_parts)
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            rounds = _check_parts(oracle, e, data, bounds,
                                  lambda b, lo, hi: b.add_synthetic_part(n, 4100, lo, hi, seed=SEED))
            assert rounds == 1                      # random data: every halo had re-synchronised
        _check_parts(oracle, e, data, split_file(n, 4),
                     lambda b, lo, hi: b.add_synthetic_part(n, 4100, lo, hi, seed=SEED), one_batch=True)
        # and the engine's own whole-file run gives the same rows
        with e.batch() as b:
            b.add_synthetic([n], [4100], seed=SEED)
            b.run()
            ref = _whole(oracle, e, data)
            assert np.array_equal(b.chunks()["sha256"], ref["sha256"])


def test_path_parts_forced_cuts_need_every_round(oracle, tmp_path):
    """All zeros, max_size not dividing the group: no halo ever re-synchronises, every part's cuts
    depend on the previous part's exit -- the exchange needs one round per boundary."""
    import makisu_amd
    n = 9 * G + 777
    data = bytes(n)
    path = tmp_path / "zeros.bin"
    path.write_bytes(data)
    with makisu_amd.Engine(max_size=100000) as e:
        bounds = [(0, 2 * G), (2 * G, 5 * G), (5 * G, 6 * G), (6 * G, n)]
        rounds = _check_parts(oracle, e, data, bounds,
                              lambda b, lo, hi: b.add_path_part(str(path), lo, hi, file_size=n))
        assert 2 <= rounds <= len(bounds) + 1


@pytest.mark.parametrize("mask_bits,min_size,max_size", [
    (4, 128, 3000), (9, 512, 10000), (16, 4096, 262144), (18, 2048, 300000), (13, 2048, 1 << 20),
    (32, 2048, 65536)])
def test_parts_param_sweep(oracle, tmp_path, mask_bits, min_size, max_size):
    """Dense tiles, chunks longer than a group (halo of several groups, or reaching back to the
    file's first byte), no candidates at all."""
    import makisu_amd
    n = 11 * G + 4321
    data = oracle.synth_fill(SEED, 4200, 0, 4 * G).tobytes() + bytes(2 * G + 99) + \
        oracle.synth_fill(SEED, 4201, 0, n - 6 * G - 99).tobytes()
    path = tmp_path / "mix.bin"
    path.write_bytes(data)
    with makisu_amd.Engine(mask_bits=mask_bits, min_size=min_size, max_size=max_size) as e:
        bounds = [(0, G), (G, 4 * G), (4 * G, 7 * G), (7 * G, 8 * G), (8 * G, n)]
        _check_parts(oracle, e, data, bounds,
                     lambda b, lo, hi: b.add_path_part(str(path), lo, hi, file_size=n))


def test_parts_next_to_ordinary_files(oracle, tmp_path):
    """A batch holding whole files and parts of two different split files."""
    import makisu_amd
    from makisu_amd.distributed import resolve_parts_local
    nA, nB = 6 * G + 100, 5 * G
    A = oracle.synth_fill(SEED, 4300, 0, nA).tobytes()
    B = oracle.synth_fill(SEED, 4301, 0, nB).tobytes()
    small = [oracle.synth_fill(SEED, 4310 + i, 0, s).tobytes() for i, s in enumerate([100, 70000, 3 * G + 5])]
    with makisu_amd.Engine() as e:
        refA, refB = _whole(oracle, e, A), _whole(oracle, e, B)
        b0, b1 = e.batch(), e.batch()
        try:
            b0.add_bytes(small[0]); b0.add_synthetic_part(nA, 4300, 0, 3 * G, seed=SEED)      # noqa: E702
            b0.add_bytes(small[1]); b0.add_synthetic_part(nB, 4301, 2 * G, nB, seed=SEED)     # noqa: E702
            b1.add_synthetic_part(nB, 4301, 0, 2 * G, seed=SEED); b1.add_bytes(small[2])      # noqa: E702
            b1.add_synthetic_part(nA, 4300, 3 * G, nA, seed=SEED)
            resolve_parts_local([(b0, [("A", 0), ("B", 1)]), (b1, [("B", 0), ("A", 1)])])
            b0.run(); b1.run()                                                                 # noqa: E702
            c0, c1 = b0.chunks().copy(), b1.chunks().copy()
            gotA = np.concatenate([c0[c0["file_index"] == 1], c1[c1["file_index"] == 2]])
            gotB = np.concatenate([c1[c1["file_index"] == 0], c0[c0["file_index"] == 3]])
            for got, ref in ((gotA, refA), (gotB, refB)):
                assert np.array_equal(got["offset"], ref["offset"])
                assert np.array_equal(got["sha256"], ref["sha256"])
            for blob, (c, f) in zip(small, ((c0, 0), (c0, 2), (c1, 1))):
                ref = _whole(oracle, e, blob)
                assert np.array_equal(c[c["file_index"] == f]["sha256"], ref["sha256"])
            # a second submit of the same batches (bench steps) gives the same rows
            b0.rerun()
            assert np.array_equal(b0.chunks()["sha256"], c0["sha256"])
        finally:
            b0.free(); b1.free()                                                               # noqa: E702


def test_part_errors(tmp_path):
    import makisu_amd
    with makisu_amd.Engine() as e:
        with e.batch() as b:
            with pytest.raises(makisu_amd.MiError):
                b.add_synthetic_part(10 * G, 1, 100, 2 * G)            # begin not aligned
            with pytest.raises(makisu_amd.MiError):
                b.add_synthetic_part(10 * G, 1, 0, G + 5)              # end neither aligned nor the file's
            with pytest.raises(makisu_amd.MiError):
                b.add_synthetic_part(10 * G, 1, 2 * G, 11 * G)         # past the file
            b.add_synthetic_part(10 * G, 1, 2 * G, 4 * G)
            with pytest.raises(makisu_amd.MiError, match="not confirmed"):
                b.run()                                                # entry never confirmed
            b.scan_cuts()
            st = b.parts()[0]
            assert st["entry_confirmed"] == 0 and 2 * G - 65536 <= st["entry"] <= 2 * G
            with pytest.raises(makisu_amd.MiError):
                b.set_part_entry(0, 2 * G - 65537)                     # farther back than max_size
            with pytest.raises(makisu_amd.MiError):
                b.set_part_entry(0, 2 * G + 1)                         # inside the part
            b.set_part_entry(0, st["entry"])
            b.fix_cuts()
            b.run()
            assert b.parts()[0]["entry_confirmed"] == 1


def test_plain_c_parts(oracle, tmp_path):
    """The parts interface from plain C (tests/cabi/parts_driver.c): the loop INTEGRATION.md gives
    a Go host, fixing each part before the next one reads its exit -- so even data that never
    re-synchronises settles in ONE sweep; the printed rows are the whole file's rows."""
    import subprocess
    import makisu_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "parts_driver")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cabi", "parts_driver.c"), "-o", exe,
                           "-L", os.path.join(root, "makisu_amd"), "-lmakisu_mi",
                           "-Wl,-rpath," + os.path.join(root, "makisu_amd")])
    cases = [(oracle.synth_fill(SEED, 4400, 0, 13 * G + 999).tobytes(), 5),
             (bytes(3 * G) + oracle.synth_fill(SEED, 4401, 0, 2 * G + 1).tobytes() + b"\x01" * (4 * G), 4),
             (oracle.synth_fill(SEED, 4402, 0, 70000).tobytes(), 3)]            # smaller than one group
    with makisu_amd.Engine() as e:
        for i, (data, n_parts) in enumerate(cases):
            path = tmp_path / ("whole%d.bin" % i)
            path.write_bytes(data)
            out = subprocess.run([exe, str(path), str(n_parts)], check=True, capture_output=True, text=True)
            rows = [l.split() for l in out.stdout.splitlines() if l.startswith("C ")]
            ref = _whole(oracle, e, data)
            assert [(int(r[1]), int(r[2]), r[3]) for r in rows] == \
                [(int(c["offset"]), int(c["length"]), "sha256:" + c["sha256"].tobytes().hex()) for c in ref]
            assert out.stdout.splitlines()[-1].startswith("R ")


def test_scan_cuts_on_a_batch_without_parts(oracle):
    """mi_batch_scan_cuts is not reserved for parts: the cuts it makes are the ones the following
    submit uses (no second Gear pass), and a later rerun makes them again -- same rows every time."""
    import makisu_amd
    blobs = [oracle.synth_fill(SEED, 4500 + i, 0, n).tobytes() for i, n in enumerate([100, 70000, 5 * G + 3, 0, 2 * G])]
    with makisu_amd.Engine() as e:
        with e.batch() as b:
            for blob in blobs:
                b.add_bytes(blob)
            b.scan_cuts()
            assert b.parts() == []
            b.fix_cuts()                                   # nothing to do
            b.run()
            first = b.chunks().copy()
            b.rerun()
            again = b.chunks().copy()
        for blob, f in zip(blobs, range(len(blobs))):
            ref = _whole(oracle, e, blob) if len(blob) else []
            mine = first[first["file_index"] == f]
            assert len(mine) == len(ref)
            if len(ref):
                assert np.array_equal(mine["offset"], ref["offset"]) and np.array_equal(mine["sha256"], ref["sha256"])
        assert np.array_equal(first["sha256"], again["sha256"]) and np.array_equal(first["offset"], again["offset"])


def _parts_rank_worker(rank, world, port, n, cid, zeros, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import makisu_amd
    from makisu_amd import distributed as mdist
    from makisu_amd.workloads import split_file
    torch.cuda.set_device(0)                       # the test box has one GPU: both ranks share it
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bounds = split_file(n, 2 * world)              # two parts per rank, interleaved over the ranks
    cfg = dict(max_size=100000) if zeros else {}
    with makisu_amd.Engine(device=0, **cfg) as eng, eng.batch() as b:
        keys = []
        for k, (lo, hi) in enumerate(bounds):
            if k % world == rank:
                if zeros:
                    b.add_path_part(zeros, lo, hi, file_size=n)
                else:
                    b.add_synthetic_part(n, cid, lo, hi, seed=SEED)
                keys.append((0, k))
        rounds = mdist.resolve_parts(b, keys)
        b.run()
        ch = b.chunks().copy()
        rows = [(k[1], ch[ch["file_index"] == i]["offset"].tolist(), ch[ch["file_index"] == i]["sha256"].tobytes())
                for i, k in enumerate(keys)]
    q.put((rank, rounds, rows))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["random", "zeros"])
def test_two_ranks_resolve_parts_on_gpu(oracle, tmp_path, kind):
    """distributed.resolve_parts through the real engine: two processes (sharing the box's one GPU,
    gloo as the transport) own alternating parts of one file, exchange exits, fix, scan; the rows of
    all parts in part order are the oracle's rows of the whole file.  'zeros': forced cuts out of
    phase with the groups -- the exchange has to run one round per boundary."""
    import socket
    import torch.multiprocessing as mp
    import makisu_amd
    n = 9 * G + 4321
    if kind == "zeros":
        data = bytes(n)
        path = tmp_path / "z.bin"
        path.write_bytes(data)
        zeros, cfg = str(path), dict(max_size=100000)
    else:
        data = oracle.synth_fill(SEED, 4600, 0, n).tobytes()
        zeros, cfg = None, {}
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()     # noqa: E702
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_parts_rank_worker, args=(r, 2, port, n, 4600, zeros, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    parts = sorted(x for _, _, rows in res for x in rows)
    offsets = [o for _, offs, _ in parts for o in offs]
    digests = b"".join(d for _, _, d in parts)
    with makisu_amd.Engine(**cfg) as e:
        ref = _whole(oracle, e, data)
    assert offsets == ref["offset"].tolist()
    assert digests == ref["sha256"].tobytes()
    rounds = {r for _, r, _ in res}
    assert len(rounds) == 1 and (rounds == {1} if kind == "random" else rounds.pop() >= 3)


@pytest.mark.parametrize("seed", range(8))
def test_parts_random_cases(oracle, tmp_path, seed):
    """Seeded random cases: parameters, content made of random / zero / periodic stretches, part
    bounds on random group boundaries, parts in separate batches or one."""
    import makisu_amd
    rng = np.random.default_rng(1000 + seed)
    mask_bits, min_size, max_size = [(13, 2048, 65536), (11, 512, 8192), (8, 256, 4096), (15, 4096, 200000),
                                     (13, 2048, 100000), (20, 2048, 524288), (6, 128, 1024), (12, 1024, 262144)][seed]
    pieces = []
    for _ in range(int(rng.integers(2, 6))):
        kind, n = int(rng.integers(0, 3)), int(rng.integers(1, 5 * G))
        if kind == 0:
            pieces.append(oracle.synth_fill(SEED, 5000 + int(rng.integers(0, 1000)), 0, n).tobytes())
        elif kind == 1:
            pieces.append(bytes(n))
        else:
            period = oracle.synth_fill(SEED, 6000 + seed, 0, int(rng.integers(100, 9000))).tobytes()
            pieces.append((period * (n // len(period) + 1))[:n])
    data = b"".join(pieces)
    n = len(data)
    groups = -(-n // G)
    k = int(rng.integers(1, min(6, groups) + 1))
    cuts = sorted(set(int(x) for x in rng.integers(1, groups, k - 1))) if groups > 1 and k > 1 else []
    edges = [0] + [c * G for c in cuts] + [n]
    bounds = [(a, b) for a, b in zip(edges, edges[1:]) if b > a]
    path = tmp_path / "case.bin"
    path.write_bytes(data)
    with makisu_amd.Engine(mask_bits=mask_bits, min_size=min_size, max_size=max_size) as e:
        _check_parts(oracle, e, data, bounds, lambda b, lo, hi: b.add_path_part(str(path), lo, hi, file_size=n),
                     one_batch=bool(seed % 2))


def test_parts_of_a_3gib_file_equal_the_whole(oracle):
    """Full-size parts: a 3 GiB synthetic file whole (one batch) and as three 1 GiB parts (a batch
    each): identical chunk rows (offsets, lengths, digests), and a sample of the digests re-hashed
    on the host from the generator."""
    import hashlib
    import makisu_amd
    from makisu_amd.distributed import resolve_parts_local
    from makisu_amd.workloads import split_file
    n, cid = 3 * (1 << 30) + 12345, 4700
    with makisu_amd.Engine() as e:
        with e.batch() as b:
            b.add_synthetic([n], [cid], seed=SEED)
            b.run()
            whole = b.chunks().copy()
            whole_root = bytes(b.files()["chunk_root"][0])
        batches = [e.batch() for _ in range(3)]
        try:
            owners = []
            for k, (b, (lo, hi)) in enumerate(zip(batches, split_file(n, 3))):
                b.add_synthetic_part(n, cid, lo, hi, seed=SEED)
                owners.append((b, [(0, k)]))
            assert resolve_parts_local(owners) == 1
            rows = []
            for b in batches:
                b.run()
                rows.append(b.chunks().copy())
            got = np.concatenate(rows)
        finally:
            for b in batches:
                b.free()
    assert len(got) == len(whole)
    assert np.array_equal(got["offset"], whole["offset"]) and np.array_equal(got["length"], whole["length"])
    assert np.array_equal(got["sha256"], whole["sha256"])
    assert makisu_amd.chunk_root(got["sha256"]) == whole_root       # the parts' digests give the file's root
    rng = np.random.default_rng(4)
    for i in rng.integers(0, len(got), 40):
        r = got[int(i)]
        off, ln = int(r["offset"]), int(r["length"])
        blob = oracle.synth_fill(SEED, cid, off, ln)
        assert bytes(r["sha256"]).hex() == hashlib.sha256(blob.tobytes()).hexdigest()
