"""BASELINE.json's synthetic configs as (sizes, content ids, seed) lists per rank.

SURVEY.md 8(d): content = counter-mode splitmix64 keyed by (seed, content id, offset / 8),
generated ON THE DEVICE by mi_batch_add_synthetic; equal content ids = byte-identical files.
Deterministic: every rank builds the same global list and takes its shard, so nothing has to
be communicated to agree on the workload.  Used by bench.py and by the parity tests (which
regenerate the same files with the oracle's generator).  Nothing here imports oracle/.

  C2  100 000 x 64 KiB per GPU, distinct contents (seed S)
  C3  1 000 x 128 MiB per GPU (seed S + 1)
  C4  N x 1 250 000 x 64 KiB, global file index i -> rank i mod N (seed S)
  C5  sizes Zipf(s = 1.1) over the 21 log2 buckets 2^10 .. 2^30 B (bucket rank 1 = 1 KiB; uniform
      inside a bucket, the last bucket is exactly 1 GiB), fixed total per GPU, 90 % of the files (by
      count) are copies of the other 10 % (seeded choice); longest-processing-time sharding by bytes,
      files >= 256 MiB split into one part per GPU when there are several (seed S + 2)
  C5U the round-1/2 stand-in: sizes 2^U(10, 30) (log-uniform: a large-file workload), same copies
"""
import numpy as np

SEED = 0x4D414B49
KIB, MIB, GIB = 1 << 10, 1 << 20, 1 << 30


def shard_round_robin(n_files, rank, world):
    """C4: file_index mod world."""
    return np.arange(rank, n_files, world, dtype=np.int64)


def shard_lpt(sizes, world):
    """C5: greedy longest-processing-time by bytes.  Returns a list of index arrays, one
    per rank; deterministic (ties broken by lower rank, files visited largest first,
    stable for equal sizes)."""
    import heapq
    sizes = np.asarray(sizes, dtype=np.int64)
    order = np.argsort(-sizes, kind="stable")
    heap = [(0, r) for r in range(world)]          # (load, rank): lowest load, then lowest rank
    out = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(int(i))
        heapq.heappush(heap, (load + int(sizes[i]), r))
    return [np.array(sorted(x), dtype=np.int64) for x in out]


class Shard:
    """One rank's share of a config: parallel arrays + what the generator knows in closed form."""

    def __init__(self, name, seed, sizes, cids, global_index, n_global_files, originals=None,
                 describe=""):
        self.name = name
        self.seed = int(seed)
        self.sizes = np.ascontiguousarray(sizes, dtype=np.uint64)
        self.cids = np.ascontiguousarray(cids, dtype=np.uint64)
        self.global_index = np.ascontiguousarray(global_index, dtype=np.int64)
        self.n_global_files = int(n_global_files)
        # originals[i]: file i is the job-wide FIRST file with its content id (global file order).
        # sum of n_chunks over the originals of all ranks = the job's unique-chunk count (contents
        # are independent random streams: no chunk repeats across or inside distinct contents)
        self.originals = (np.ones(len(self.sizes), dtype=bool) if originals is None
                          else np.ascontiguousarray(originals, dtype=bool))
        self.describe = describe
        self.parts = None          # c5 with several ranks: per item (file_size, begin, end, part_no)
        self.imbalance = 1.0

    @property
    def n_files(self):
        return len(self.sizes)

    @property
    def n_bytes(self):
        return int(self.sizes.sum())


def c2(rank=0, world=1, files_per_gpu=100000, generation=0):
    """generation: which of the in-flight batches (distinct content per batch)."""
    n = files_per_gpu * world
    idx = shard_round_robin(n, rank, world)
    cids = idx + generation * n
    return Shard("c2", SEED, np.full(len(idx), 64 * KIB), cids, idx, n,
                 describe="C2: %d x 64 KiB synthetic files per GPU" % files_per_gpu)


def c3(rank=0, world=1, files_per_gpu=1000, file_bytes=128 * MIB, generation=0):
    n = files_per_gpu * world
    idx = shard_round_robin(n, rank, world)
    return Shard("c3", SEED + 1, np.full(len(idx), file_bytes), idx + generation * n, idx, n,
                 describe="C3: %d x %d MiB synthetic files per GPU" % (files_per_gpu, file_bytes // MIB))


def c4(rank=0, world=8, files_per_gpu=1250000, generation=0):
    n = files_per_gpu * world                       # 10 M at 8 GPUs
    idx = shard_round_robin(n, rank, world)
    return Shard("c4", SEED, np.full(len(idx), 64 * KIB), idx + generation * n, idx, n,
                 describe="C4: %d x 64 KiB files, file index mod %d" % (n, world))


ZIPF_S = 1.1


def zipf_bucket_probs(lo_log2=10, hi_log2=30, s=ZIPF_S):
    """P(bucket b), b = lo_log2 .. hi_log2: Zipf over the bucket RANK (1 = the smallest sizes)."""
    r = np.arange(1, hi_log2 - lo_log2 + 2, dtype=np.float64)
    w = r ** -s
    return w / w.sum()


def c5_global(world=1, bytes_per_gpu=64 * GIB, lo_log2=10, hi_log2=30, law="zipf"):
    """The job-wide C5 file list: (sizes, cids, originals).  law "zipf": BASELINE.json configs[4] /
    SURVEY.md 8(d) -- by COUNT three files in four are below 128 KiB, by BYTES nearly everything sits
    in the files of 64 MiB and more; "loguniform": 2^U(lo, hi), rounds 1-2."""
    rng = np.random.default_rng(SEED + 2)
    target = bytes_per_gpu * world
    # the 10 % distinct contents: sizes drawn until they hold a tenth of the bytes; then nine copies
    # per original on average, drawn with replacement
    probs = zipf_bucket_probs(lo_log2, hi_log2)
    sizes = []
    acc = 0
    while acc < target // 10:
        if law == "zipf":
            b = lo_log2 + int(rng.choice(len(probs), p=probs))
            s = (1 << b) if b == hi_log2 else int(rng.integers(1 << b, 1 << (b + 1)))
        else:
            s = int(2.0 ** rng.uniform(lo_log2, hi_log2))
        sizes.append(s)
        acc += s
    distinct = len(sizes)
    sizes = np.array(sizes, dtype=np.int64)
    src = rng.integers(0, distinct, 9 * distinct)
    all_sizes = np.concatenate([sizes, sizes[src]])
    cids = np.concatenate([np.arange(distinct), src])
    keep = np.cumsum(all_sizes) <= target
    keep[:distinct] = True
    all_sizes, cids = all_sizes[keep], cids[keep]
    originals = np.zeros(len(cids), dtype=bool)
    originals[:distinct] = True                     # copies come after every original
    return all_sizes, cids, originals


def c5(rank=0, world=1, bytes_per_gpu=64 * GIB, generation=0, lo_log2=10, hi_log2=30, law="zipf",
       split_threshold=256 * MIB):
    """One rank's C5 items.  With several ranks, files of split_threshold bytes and more are cut into
    one part per rank (plan_split), everything else goes whole to the least loaded
    rank.  shard.parts (or None): per item (file_size, begin, end, part_no) -- whole files have
    part_no -1; shard.imbalance = max / mean bytes per rank."""
    sizes, cids, originals = c5_global(world, bytes_per_gpu, lo_log2, hi_log2, law)
    n_contents = int(cids.max()) + 1
    name = "c5" if law == "zipf" else "c5u"
    what = ("Zipf s=%.1f over log2 buckets 2^%d..2^%d B" % (ZIPF_S, lo_log2, hi_log2)) if law == "zipf" \
        else "sizes 2^U(%d,%d) B" % (lo_log2, hi_log2)
    items, ranks = plan_split(sizes, world, split_threshold)
    loads = np.zeros(world, dtype=np.int64)
    for (f, pno, b, e), r in zip(items, ranks):
        loads[r] += e - b
    mine = sorted((it for it, r in zip(items, ranks) if r == rank), key=lambda it: (it[0], it[2]))
    files = np.array([it[0] for it in mine], dtype=np.int64)
    isz = np.array([it[3] - it[2] for it in mine], dtype=np.int64)
    sh = Shard(name, SEED + 2, isz, cids[files] + generation * n_contents, files, len(sizes),
               originals=originals[files],
               describe="C5: %s, 90 %% of the files copies of the other 10 %%: %d files / %d distinct contents "
                        "job-wide, LPT shards%s" % (what, len(sizes), n_contents,
                                                    ", files >= %d MiB as %d parts" % (split_threshold // MIB, world)
                                                    if world > 1 else ""))
    n_split = sum(1 for it in mine if it[1] is not None)
    sh.parts = [(int(sizes[it[0]]), it[2], it[3], -1 if it[1] is None else it[1]) for it in mine] if n_split else None
    sh.imbalance = float(loads.max() / max(1.0, loads.mean()))
    sh.n_split_files_job = int((sizes >= split_threshold).sum()) if world > 1 else 0
    return sh


def c5u(rank=0, world=1, bytes_per_gpu=32 * GIB, generation=0, lo_log2=10, hi_log2=30):
    return c5(rank, world, bytes_per_gpu, generation, lo_log2, hi_log2, law="loguniform")


CONFIGS = {"c2": c2, "c3": c3, "c4": c4, "c5": c5, "c5u": c5u}


def batch_bytes_hint(sh):
    """Arena bytes a batch of this shard needs (files on 256-byte boundaries, halos of the parts)."""
    return sh.n_bytes + (sh.n_files + 8) * 4096 + (len(sh.parts or ()) << 19)


def fill_batch(b, sh):
    """Adds a shard's items to a batch: runs of whole files through add_synthetic, the parts of split files
    (c5 on several GPUs) through add_synthetic_part.  Returns the keys of the parts in the order b.parts()
    lists them -- (file key, part number), what resolve_parts / resolve_parts_local exchange -- [] without."""
    if sh.parts is None:
        b.add_synthetic(sh.sizes, sh.cids, seed=sh.seed)
        return []
    keys, i, n = [], 0, sh.n_files
    while i < n:
        if sh.parts[i][3] < 0:
            j = i
            while j < n and sh.parts[j][3] < 0:
                j += 1
            b.add_synthetic(sh.sizes[i:j], sh.cids[i:j], seed=sh.seed)
            i = j
        else:
            fsize, begin, end, pno = sh.parts[i]
            b.add_synthetic_part(fsize, int(sh.cids[i]), begin, end, seed=sh.seed)
            keys.append((int(sh.cids[i]) * 4 * sh.n_global_files + int(sh.global_index[i]), pno))
            i += 1
    return keys


PART_ALIGN = 256 * KIB          # MI_PART_ALIGN: part bounds are multiples of the 256 KiB CDC group


def split_file(file_size, n_parts, align=PART_ALIGN):
    """SURVEY.md 8(e) "files >= 256 MiB split ... across GPUs": [begin, end) of n_parts (or fewer)
    nearly equal parts with group-aligned bounds; the last part ends at file_size."""
    groups = -(-int(file_size) // align)
    n_parts = max(1, min(int(n_parts), groups))
    cuts = [(groups * i // n_parts) * align for i in range(n_parts)] + [int(file_size)]
    return [(cuts[i], cuts[i + 1]) for i in range(n_parts) if cuts[i + 1] > cuts[i]]


def plan_split(sizes, world, threshold=256 * MIB):
    """Work items for `world` GPUs: files below `threshold` stay whole, larger ones become `world`
    parts.  Returns a list of (file, part_no, begin, end) -- part_no is None for a whole file --
    and the rank of each item: parts go to consecutive ranks, whole files to the least loaded
    rank (longest first), like shard_lpt."""
    import heapq
    sizes = np.asarray(sizes, dtype=np.int64)
    items, ranks = [], []
    load = [0] * world
    for f in np.argsort(-sizes, kind="stable"):
        if sizes[f] >= threshold and world > 1:
            for k, (b, e) in enumerate(split_file(sizes[f], world)):
                items.append((int(f), k, b, e))
                ranks.append(k % world)
                load[k % world] += e - b
    heap = [(load[r], r) for r in range(world)]
    heapq.heapify(heap)
    for f in np.argsort(-sizes, kind="stable"):
        if sizes[f] >= threshold and world > 1:
            continue
        ld, r = heapq.heappop(heap)
        items.append((int(f), None, 0, int(sizes[f])))
        ranks.append(r)
        heapq.heappush(heap, (ld + int(sizes[f]), r))
    return items, ranks
