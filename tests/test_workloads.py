"""CPU tests of BASELINE.json's config generators (makisu_amd/workloads.py): every rank must derive
the same job from nothing but (config, rank, world), and the closed forms bench.py checks the GPU
against must hold by construction."""
import os
import numpy as np

from makisu_amd import workloads as W
from makisu_amd import distributed as D


def test_c4_round_robin_partition():
    world, per = 8, 1000
    shards = [W.c4(r, world, per) for r in range(world)]
    allidx = np.concatenate([s.global_index for s in shards])
    assert sorted(allidx.tolist()) == list(range(world * per))                 # a partition
    for r, s in enumerate(shards):
        assert (s.global_index % world == r).all()                             # file index mod N
        assert s.n_files == per and (s.sizes == 65536).all()
        assert np.array_equal(s.cids, s.global_index.astype(np.uint64))        # distinct content per file
        assert s.originals.all()
    assert W.c4(3, 8).n_files == 1250000 and W.c4(3, 8).n_global_files == 10_000_000   # BASELINE configs[3]
    gen1 = W.c4(0, 8, per, generation=1)
    assert len(set(gen1.cids.tolist()) & set(shards[0].cids.tolist())) == 0    # batches in flight differ


def test_c5_lpt_shards_are_a_balanced_partition_with_known_duplicates():
    world = 4
    sizes, cids, originals = W.c5_global(world, bytes_per_gpu=W.GIB, hi_log2=26)
    assert (sizes >= 1 << 10).all() and (sizes <= 1 << 26).all()
    n_contents = int(cids.max()) + 1
    assert originals[:n_contents].all() and not originals[n_contents:].any()
    assert len(sizes) - n_contents > 5 * n_contents                            # copies dominate by count
    assert np.array_equal(sizes[n_contents:], sizes[cids[n_contents:]])        # a copy has its original's size
    shards = [W.c5(r, world, bytes_per_gpu=W.GIB, hi_log2=26) for r in range(world)]
    allidx = np.concatenate([s.global_index for s in shards])
    assert sorted(allidx.tolist()) == list(range(len(sizes)))
    loads = [s.n_bytes for s in shards]
    assert max(loads) - min(loads) <= int(sizes.max())                         # LPT: within one file
    assert sum(int(s.originals.sum()) for s in shards) == n_contents           # every original lives on one rank
    again = W.c5(2, world, bytes_per_gpu=W.GIB, hi_log2=26)                    # deterministic
    assert np.array_equal(again.sizes, shards[2].sizes) and np.array_equal(again.cids, shards[2].cids)


def test_shard_lpt_matches_the_reference_form():
    rng = np.random.default_rng(1)
    sizes = rng.integers(1, 10**6, 500)
    got = W.shard_lpt(sizes, 5)
    # the O(n * world) form used in round 1: lowest load, ties to the lowest rank
    order = np.argsort(-sizes, kind="stable")
    load = np.zeros(5, dtype=np.int64)
    want = [[] for _ in range(5)]
    for i in order:
        r = int(np.argmin(load))
        want[r].append(int(i))
        load[r] += int(sizes[i])
    assert [g.tolist() for g in got] == [sorted(w) for w in want]
    assert D.shard_lpt is W.shard_lpt and D.shard_round_robin is W.shard_round_robin


def test_c2_c3_shapes():
    s = W.c2(0, 1)
    assert s.n_files == 100000 and s.n_bytes == 100000 * 65536 and s.seed == W.SEED
    s = W.c3(1, 2, 10)
    assert s.n_files == 10 and (s.sizes == 128 * W.MIB).all() and s.seed == W.SEED + 1
    assert (s.global_index % 2 == 1).all()


def test_c5_is_zipf_over_log2_buckets():
    """BASELINE.json configs[4] / SURVEY.md 8(d): sizes Zipf(s = 1.1) over the 21 log2 buckets
    2^10 .. 2^30 (rank 1 = the 1 KiB bucket), uniform inside a bucket, the last bucket exactly 1 GiB;
    90 % of the files (by count) copies of the other 10 %."""
    p = W.zipf_bucket_probs()
    assert len(p) == 21 and abs(p.sum() - 1) < 1e-12
    assert np.allclose(p[0] / p[1], 2 ** 1.1) and np.allclose(p[0] / p[20], 21 ** 1.1)
    sizes, cids, originals = W.c5_global(8, bytes_per_gpu=64 * W.GIB)
    assert sizes.min() >= 1 << 10 and sizes.max() == 1 << 30
    n_contents = int(originals.sum())
    assert 0.85 <= 1 - n_contents / len(sizes) <= 0.91                       # ~90 % copies by count
    assert 0.8 * 512 * W.GIB <= sizes.sum() <= 512 * W.GIB
    # the empirical bucket histogram of the originals follows the law (chi-square-ish, loose)
    b = np.floor(np.log2(sizes[:n_contents])).astype(int) - 10
    hist = np.bincount(b, minlength=21) / n_contents
    assert np.abs(hist - p).max() < 0.03
    assert (sizes <= 64 * W.KIB).mean() > 0.6                                # the small-file head, by count
    assert sizes[sizes >= 64 * W.MIB].sum() > 0.8 * sizes.sum()              # the large-file tail, by bytes
    # the rounds-1/2 stand-in is still there under its own name
    u = W.c5u(0, 1, bytes_per_gpu=W.GIB)
    assert u.name == "c5u" and "2^U" in u.describe


def test_c5_several_ranks_split_the_large_files_into_parts():
    """files >= 256 MiB become one part per GPU (group-aligned bounds), the rest is
    LPT-balanced; every byte of every file belongs to exactly one item of one rank."""
    world = 4
    sizes, cids, originals = W.c5_global(world, bytes_per_gpu=8 * W.GIB)
    shards = [W.c5(r, world, bytes_per_gpu=8 * W.GIB) for r in range(world)]
    covered = np.zeros(len(sizes), dtype=np.int64)
    n_parts = 0
    for sh in shards:
        assert sh.parts is not None and sh.imbalance < 1.01
        for i, (fsize, b, e, pno) in enumerate(sh.parts):
            f = int(sh.global_index[i])
            assert fsize == sizes[f] and 0 <= b < e <= fsize and int(sh.sizes[i]) == e - b
            if pno < 0:
                assert (b, e) == (0, fsize) and fsize < 256 * W.MIB
            else:
                n_parts += 1
                assert fsize >= 256 * W.MIB and b % W.PART_ALIGN == 0 and (e % W.PART_ALIGN == 0 or e == fsize)
            covered[f] += e - b
    assert np.array_equal(covered, sizes)
    assert n_parts == world * int((sizes >= 256 * W.MIB).sum()) > 0
    loads = [s.n_bytes for s in shards]
    assert max(loads) / (sum(loads) / world) == shards[0].imbalance


def test_bench_default_batches_in_flight():
    """bench.py: two batches in flight (the steady state `value` is quoted on since round 4, as in rounds 1-2; the
    roofline comes from the one-batch-at-a-time steps of the same run), three for the small config with an exchange."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert [bench.default_inflight(c, False) for c in ("c2", "c3", "c4", "c5", "c5u")] == [2, 1, 2, 2, 2]
    assert bench.default_inflight("c2", True) == 3 and bench.default_inflight("c4", True) == 2
    assert bench.default_inflight("c5", True) == 2


def test_fill_batch_hands_over_whole_files_in_runs_and_parts_one_by_one():
    """workloads.fill_batch (bench.py and the exchange tests add a shard's items through it): runs of whole files go through
    add_synthetic in shard order, every part of a split file through add_synthetic_part, and the returned keys name the
    parts in the order the batch lists them -- (file key, part number), equal on every rank that holds a part of that file."""
    class FakeBatch:
        def __init__(self):
            self.calls = []

        def add_synthetic(self, sizes, cids, seed):
            self.calls.append(("files", [int(x) for x in sizes], [int(x) for x in cids], seed))

        def add_synthetic_part(self, fsize, cid, begin, end, seed):
            self.calls.append(("part", int(fsize), int(cid), int(begin), int(end), seed))

    world = 4
    shards = [W.c5(r, world, bytes_per_gpu=2 * W.GIB, split_threshold=32 * W.MIB) for r in range(world)]
    all_keys = {}
    for r, sh in enumerate(shards):
        fb = FakeBatch()
        keys = W.fill_batch(fb, sh)
        n_parts = sum(1 for p in sh.parts if p[3] >= 0)
        assert len(keys) == n_parts == sum(1 for c in fb.calls if c[0] == "part") > 0
        handed = []
        for c in fb.calls:                                       # the items in shard order, whichever call carried them
            handed += [("f", n) for n in c[1]] if c[0] == "files" else [("p", c[4] - c[3])]
        assert [n for _, n in handed] == [int(x) for x in sh.sizes]
        assert [k == "p" for k, _ in handed] == [p[3] >= 0 for p in sh.parts]
        assert W.batch_bytes_hint(sh) >= sh.n_bytes + len(sh.parts) * 262144       # room for the parts' halos
        for key in keys:
            all_keys.setdefault(key[0], []).append((key[1], r))
    for fkey, parts in all_keys.items():                         # every split file: parts 0 .. world-1, one per rank
        assert sorted(p for p, _ in parts) == list(range(world)) and len({rk for _, rk in parts}) == world
    whole = W.c4(1, 8, 1000)
    fb = FakeBatch()
    assert W.fill_batch(fb, whole) == [] and len(fb.calls) == 1 and fb.calls[0][0] == "files"


def test_bench_closed_form_check_tolerates_only_one_byte_tail_coincidences():
    """bench.py's dedup_check: the job-wide unique count may fall short of the generator's closed form by the handful of
    equal 1-byte tail chunks distinct contents can share (2 + closed_form / 50 000), never exceed it, never be missing"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ok = lambda got, want: bench.closed_form_check(got, want)["ok"]            # noqa: E731
    assert ok(721319, 721319) and ok(9023949, 9023970) and ok(100, 102)
    assert not ok(103, 102) and not ok(99, 102) and not ok(None, 5) and not ok(9023000, 9023970)
    assert bench.closed_form_check(9023949, 9023970)["short_chunk_coincidences"] == 21
