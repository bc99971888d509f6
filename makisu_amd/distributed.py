"""Multi-GPU sharding and the chunk-digest exchange.

One process per GPU.  Files are independent units, so the scan itself needs no
collective: every rank scans its shard.  The ONE exchange step is the all-gather
of the per-rank chunk-digest arrays (32 B per chunk) so every rank can mark
duplicates against the global set -- the chunk-granular analogue of the
reference's content-addressed layer dedup (lib/builder/step/common.go:88-91).

`torch.distributed` is plumbing here (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests); the marking itself is the HIP kernel behind
`mi_dedup_mark`.  Nothing in this file imports oracle/.
"""
import numpy as np
import torch
import torch.distributed as dist


from .workloads import shard_lpt, shard_round_robin  # noqa: E402,F401  (re-exported)


class DeviceArray:
    """Zero-copy view of engine-owned device memory for torch (__cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def digests_tensor(batch, device):
    """The batch's n_chunks x 32 digest array as a torch uint8 tensor on `device`."""
    ptr, n = batch.device_digests()
    if n == 0:
        return torch.empty((0, 32), dtype=torch.uint8, device=device)
    return torch.as_tensor(DeviceArray(ptr, (n, 32), "|u1"), device=device)


_host_groups = {}
_exchange_streams = {}


def _host_group(group):
    """A process group for the HOST-side scalars of the exchange (per-rank counts, the unique
    count).  On a GPU busy with persistent scan kernels even an 8-byte device collective waits
    milliseconds for a hardware queue slot, and its result needs a host sync on top; the counts
    are host integers anyway, so they travel over a gloo group (TCP/shared memory on one node)
    and never touch the GPU.  With a gloo default group (the CPU tests) that group is used."""
    key = id(group) if group is not None else None
    if key not in _host_groups:
        if dist.get_backend(group) == "gloo":
            _host_groups[key] = group
        else:
            ranks = None if group is None else dist.get_process_group_ranks(group)
            _host_groups[key] = dist.new_group(ranks=ranks, backend="gloo")   # collective: every rank gets here
    return _host_groups[key]


def _exchange_stream(device):
    """The torch stream the slab all-gather runs on, so that waiting for the exchange never means
    waiting for torch's default stream or for the device.  Default priority on purpose: creating
    streams with an explicit priority changed how the ROCm 7.0 runtime spreads the batch streams
    over the hardware queues and cost the batches their overlap."""
    key = (device.type, device.index)
    if key not in _exchange_streams:
        _exchange_streams[key] = torch.cuda.Stream(device=device)
    return _exchange_streams[key]


def all_gather_counts(n_local, group=None):
    """Every rank's row count, as host integers (no GPU work)."""
    world = dist.get_world_size(group)
    mine = torch.tensor([int(n_local)], dtype=torch.int64)
    out = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(out, mine, group=_host_group(group))
    return [int(t.item()) for t in out]


def all_gather_digests(local, group=None):
    """Variable-length all-gather of (n_r, 32) uint8 digest arrays.

    Step 1: all-gather the counts on the host.  Step 2: pad to the max count and all-gather the
    slabs (one collective, every xGMI link carries one peer's slab), then drop the padding.
    Returns (global (N, 32) tensor ordered by rank, counts list, first_global index of this
    rank's rows).  On a GPU the device work runs on a stream of its own that has been
    synchronised when this returns."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local.device
    counts = all_gather_counts(local.shape[0], group)
    m = max(counts) if counts else 0
    if m == 0:
        return torch.empty((0, 32), dtype=torch.uint8, device=dev), counts, 0

    def gather():
        slab = torch.zeros((m, 32), dtype=torch.uint8, device=dev)
        slab[: local.shape[0]] = local
        gathered = torch.empty((world * m, 32), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, slab, group=group)
        if all(c == m for c in counts):
            return gathered
        return torch.cat([gathered[r * m: r * m + counts[r]] for r in range(world)], dim=0).contiguous()

    if dev.type == "cuda":
        st = _exchange_stream(dev)
        with torch.cuda.stream(st):
            glob = gather()
        st.synchronize()          # only this stream: the batches in flight keep running
    else:
        glob = gather()
    return glob, counts, sum(counts[:rank])


def global_dedup(engine, batch, device, group=None, mark=None, local=None):
    """Exchange + mark.  Rewrites the batch's dup_of column with GLOBAL chunk indices
    (rank-major order).  The default marking is the HIP kernel behind mi_batch_mark_global:
    a rank only answers for its own rows (own rows build the table, rows of earlier ranks
    probe it), then the per-rank first-occurrence counts are summed on the host group.
    The gloo CPU tests pass `local` digests and inject their own checker as
    `mark(glob) -> (dup_of int64 tensor over ALL rows, n_unique)`.
    Returns (n_total, n_unique_global, first_global, dup_of) -- dup_of is a zero-copy view of
    the batch's own column (global indices) on the default path, all rows when `mark` is given."""
    if local is None:
        local = digests_tensor(batch, device)
    glob, counts, first = all_gather_digests(local, group)
    n_total = glob.shape[0]
    if mark is not None:
        dup, n_unique = mark(glob)
        return n_total, n_unique, first, dup
    n_first = batch.mark_global(glob.data_ptr(), n_total, first)     # into the batch's own column
    ptr, n_own = batch.device_dup_of()
    if device.type == "cuda" and n_own:
        dup = torch.as_tensor(DeviceArray(ptr, (n_own,), "<i8"), device=device)
    else:                                   # CPU tests drive this path with a stand-in batch
        dup = batch.dup_of_host() if n_own else torch.empty(0, dtype=torch.int64)
    t = torch.tensor([n_first], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_host_group(group))
    return n_total, int(t.item()), first, dup


# ---- parts: one file split across GPUs (include/makisu_mi.h "parts") --------------------------------
def _apply_exits(batch, keys, states, exits):
    """Gives every part of `batch` the exit of its predecessor.  keys[i] = (file_key, part_no) of the
    i-th part batch.parts() lists.  Returns True when some part's cuts have to be made again."""
    redo = False
    for (fkey, pno), st in zip(keys, states):
        if pno == 0:
            continue
        e = exits[(fkey, pno - 1)]
        if e != st["entry"]:
            redo = True
        if e != st["entry"] or not st["entry_confirmed"]:
            batch.set_part_entry(st["file_index"], e)
    return redo


def resolve_parts_local(owners):
    """All parts live in this process (several batches standing in for GPUs, or one GPU working
    through a file larger than its memory): owners = [(batch, keys), ...].  Runs the scan /
    exchange / fix rounds until every part starts at its predecessor's last cut; the batches can
    then be submitted.  Returns the number of rounds."""
    for b, _ in owners:
        b.scan_cuts()
    rounds = 0
    while True:
        rounds += 1
        states = [b.parts() for b, _ in owners]
        exits = {k: st["exit"] for (b, keys), sts in zip(owners, states) for k, st in zip(keys, sts)}
        redo = False
        for (b, keys), sts in zip(owners, states):
            if _apply_exits(b, keys, sts, exits):
                redo = True
            b.fix_cuts()
        if not redo:
            return rounds


def resolve_parts(batch, keys, group=None):
    """One process per GPU: the same rounds with the exits travelling over the host group (gloo;
    8 bytes per part and round, no device collective).  keys as in _apply_exits; every rank calls
    this, also ranks without parts (keys = [])."""
    hg = _host_group(group)
    world = dist.get_world_size(group)
    batch.scan_cuts()
    rounds = 0
    while True:
        rounds += 1
        states = batch.parts()
        counts = all_gather_counts(len(states), group)
        m = max(counts)
        redo = False
        if m:
            slab = torch.full((m, 3), -1, dtype=torch.int64)
            for i, ((fkey, pno), st) in enumerate(zip(keys, states)):
                slab[i, 0], slab[i, 1], slab[i, 2] = int(fkey), int(pno), int(st["exit"])
            out = [torch.empty((m, 3), dtype=torch.int64) for _ in range(world)]
            dist.all_gather(out, slab, group=hg)
            exits = {}
            for r in range(world):
                for row in out[r][: counts[r]].tolist():
                    exits[(row[0], row[1])] = row[2]
            redo = _apply_exits(batch, keys, states, exits)
            batch.fix_cuts()
        t = torch.tensor([1 if redo else 0], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=hg)
        if int(t.item()) == 0:
            return rounds
