"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding and the variable-length
digest all-gather + global duplicate marking plumbing (makisu_amd/distributed.py).  The marking
kernel itself needs a GPU; here the oracle stands in as the checker via the `mark` hook."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x4D414B49


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_files, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from makisu_amd import distributed as mdist
    from oracle import mi_oracle as O
    p = O.CdcParams(SEED, 13, 2048, 65536)
    # C4-style shard: file_index mod world; half of all files repeat earlier content
    mine = mdist.shard_round_robin(n_files, rank, world)
    cids = [int(i) % (n_files // 2) for i in mine]
    sizes = [20000 + 3000 * (int(i) % 5) for i in mine]          # ragged, different counts per rank
    data = np.concatenate([O.synth_fill(SEED, c, 0, s) for c, s in zip(cids, sizes)])
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    _, chunks = O.scan_batch(data, offs, sizes, p)
    local = torch.from_numpy(np.ascontiguousarray(chunks["sha256"]))

    def mark(glob):
        dup, uniq = O.dedup(glob.numpy())
        return torch.from_numpy(dup), uniq

    glob, counts, first = mdist.all_gather_digests(local)
    n_total, n_unique, first2, dup = mdist.global_dedup(None, None, torch.device("cpu"), mark=mark,
                                                        local=local)
    assert n_total == glob.shape[0] and first2 == first
    q.put((rank, [int(c) for c in counts], int(first), glob.numpy().tobytes(),
           dup.numpy().tolist(), int(n_unique), local.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_digest_all_gather_and_global_marking_world2():
    world, n_files = 2, 13
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_files, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    (r0, counts0, first0, glob0, dup0, uniq0, loc0), (r1, counts1, first1, glob1, dup1, uniq1, loc1) = res
    assert counts0 == counts1 and counts0[0] != counts0[1]        # ragged: padding path exercised
    assert first0 == 0 and first1 == counts0[0]
    assert glob0 == glob1 == loc0 + loc1                          # rank-major concatenation
    assert dup0 == dup1 and uniq0 == uniq1
    dup = np.array(dup0)
    assert (dup >= 0).sum() > 0 and uniq0 == (dup < 0).sum()
    assert (dup < np.arange(len(dup))).all()                      # always points to an earlier row


class _HostBatch:
    """Stand-in for makisu_amd.Batch on a CPU: mi_batch_mark_global's contract (own rows of the
    job-wide marking, written into the batch's column) computed by the oracle on the memory
    global_dedup hands over."""

    def __init__(self, O, n_own):
        self.O, self.n_own, self.col = O, n_own, None

    def mark_global(self, d_digests_ptr, n_total, own_first):
        import ctypes
        rows = np.ctypeslib.as_array((ctypes.c_uint8 * (n_total * 32)).from_address(d_digests_ptr))
        full, _ = self.O.dedup(rows.reshape(n_total, 32).copy())
        self.col = full[own_first: own_first + self.n_own].copy()
        self.first = own_first
        return int((self.col < 0).sum())

    def device_dup_of(self):
        return 0, self.n_own

    def dup_of_host(self):
        return torch.from_numpy(self.col)


def _worker_default_path(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from makisu_amd import distributed as mdist
    from oracle import mi_oracle as O
    rng = np.random.default_rng(100)                       # same pool on every rank
    pool = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    pick = np.random.default_rng(rank).integers(0, 50, 20 + 7 * rank)   # ragged counts, many repeats
    local = torch.from_numpy(np.ascontiguousarray(pool[pick]))
    b = _HostBatch(O, len(pick))
    n_total, n_unique, first, dup = mdist.global_dedup(None, b, torch.device("cpu"), local=local)
    assert b.first == first
    q.put((rank, n_total, n_unique, first, dup.numpy()[: len(pick)].tolist(), local.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_global_dedup_default_path_world3():
    """The production path of global_dedup (host-group counts, own-rows marking, summed unique
    count) with three ranks and ragged shards; the marking kernel is replaced by the oracle."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_default_path, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    sys.path.insert(0, ROOT)
    from oracle import mi_oracle as O
    rows = np.frombuffer(b"".join(r[5] for r in res), dtype=np.uint8).reshape(-1, 32)
    want, uniq = O.dedup(rows)
    at = 0
    for rank, n_total, n_unique, first, dup, _ in res:
        assert n_total == len(rows) and n_unique == uniq and first == at
        assert dup == want[at: at + len(dup)].tolist()
        at += len(dup)
    assert at == len(rows)


def test_shard_helpers():
    sys.path.insert(0, ROOT)
    from makisu_amd import distributed as mdist
    parts = [mdist.shard_round_robin(10, r, 4) for r in range(4)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(10))
    assert parts[1].tolist() == [1, 5, 9]
    rng = np.random.default_rng(2)
    sizes = (2 ** rng.uniform(10, 30, 400)).astype(np.int64)      # C5-like: 1 KiB .. 1 GiB
    shards = mdist.shard_lpt(sizes, 8)
    assert sorted(np.concatenate(shards).tolist()) == list(range(400))
    loads = np.array([sizes[s].sum() for s in shards])
    assert loads.max() <= loads.mean() + sizes.max()              # LPT bound
    assert [s.tolist() for s in mdist.shard_lpt(sizes, 8)] == [s.tolist() for s in shards]
