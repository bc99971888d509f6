"""Split files (include/makisu_mi.h "parts") on CPU: the planning helpers and the exit/entry
exchange of makisu_amd/distributed.py over world_size-2/3 gloo groups.  The engine's kernels need a
GPU (tests/test_gpu_parts.py); here a stand-in batch answers scan_cuts / parts / set_part_entry /
fix_cuts with the oracle's sequential chunker, so what is tested is the protocol: every part ends
up starting at its predecessor's last cut, in as many rounds as the data needs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x4D414B49
G = 256 * 1024


def test_split_file_bounds():
    from makisu_amd.workloads import split_file, PART_ALIGN
    assert PART_ALIGN == G
    for size, n in [(10 * G + 5, 3), (G, 4), (G + 1, 2), (1 << 30, 8), (7, 3), (100 * G, 7)]:
        parts = split_file(size, n)
        assert parts[0][0] == 0 and parts[-1][1] == size and len(parts) <= n
        for (b0, e0), (b1, e1) in zip(parts, parts[1:]):
            assert e0 == b1 and b1 % G == 0 and e0 > b0
        lens = [e - b for b, e in parts]
        assert max(lens) - min(lens) <= 2 * G or len(parts) == 1


def test_plan_split_balances_and_keeps_small_files_whole():
    from makisu_amd.workloads import plan_split, MIB
    sizes = [900 * MIB, 10 * MIB, 300 * MIB, 5 * MIB, 200 * MIB, 1]
    items, ranks = plan_split(sizes, 4)
    whole = [(f, b, e) for (f, k, b, e) in items if k is None]
    assert sorted(f for f, _, _ in whole) == [1, 3, 4, 5]
    for f in (0, 2):
        mine = sorted((k, b, e, r) for (ff, k, b, e), r in zip(items, ranks) if ff == f)
        assert [k for k, *_ in mine] == list(range(len(mine))) and mine[0][1] == 0 and mine[-1][2] == sizes[f]
        assert [r for *_, r in mine] == list(range(len(mine)))            # consecutive ranks
    load = [0] * 4
    for (f, k, b, e), r in zip(items, ranks):
        load[r] += e - b
    assert max(load) - min(load) <= 200 * MIB
    items1, ranks1 = plan_split(sizes, 1)                                   # one GPU: nothing is split
    assert all(k is None for _, k, _, _ in items1) and set(ranks1) == {0}


class ModelBatch:
    """What the engine does for parts, restated with the oracle's classic chunker."""

    def __init__(self, oracle, data, params):
        self.O, self.data, self.p = oracle, data, params
        self.recs = []
        self.fixes = 0

    def add_part(self, begin, end):
        halo = 0
        if begin:
            halo = -(-self.p.max_size // G) * G
            halo = min(halo, begin)
        self.recs.append(dict(file_index=len(self.recs), begin=begin, end=end, halo=halo, entry=None,
                              exit=None, entry_confirmed=int(begin == 0), set=None))

    def _cuts_from(self, entry, end):
        n = len(self.data)
        stop = min(n, end + self.p.max_size)
        ends = entry + np.asarray(self.O.cdc_classic(self.data[entry:stop], self.p), dtype=np.int64)
        if end < n:
            ends = ends[ends <= end]
        return ends

    def _select(self, r, entry):
        cuts = self._cuts_from(entry, r["end"])
        own = cuts[cuts > r["begin"]]
        r["cuts"] = own
        r["entry"] = entry
        r["exit"] = int(own[-1]) if len(own) else entry

    def scan_cuts(self):
        for r in self.recs:
            if r["begin"] == 0:
                self._select(r, 0)
                continue
            base = r["begin"] - r["halo"]
            pre = self._cuts_from(base, r["begin"])
            pre = pre[pre <= r["begin"]]
            self._select(r, int(pre[-1]) if len(pre) else base)
        return self

    def parts(self):
        return [dict(r, entry=r["set"] if r["set"] is not None else r["entry"]) for r in self.recs]

    def set_part_entry(self, file_index, entry):
        r = self.recs[file_index]
        assert r["begin"] - self.p.max_size <= entry <= r["begin"]
        r["set"] = entry
        r["entry_confirmed"] = 1

    def fix_cuts(self):
        for r in self.recs:
            if r["set"] is not None and r["set"] != r["entry"]:
                self._select(r, r["set"])
                self.fixes += 1
        return self


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _case(O, name):
    from makisu_amd.workloads import split_file
    if name == "random":
        data = O.synth_fill(SEED, 77, 0, 9 * G + 1234)
        p = O.CdcParams(SEED, 13, 2048, 65536)
    else:                                   # forced cuts only, out of phase with the groups
        data = np.zeros(9 * G + 1234, dtype=np.uint8)
        p = O.CdcParams(SEED, 13, 2048, 100000)
    return data, p, split_file(len(data), 5)


def _worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from makisu_amd import distributed as mdist
    from oracle import mi_oracle as O
    data, p, bounds = _case(O, name)
    b = ModelBatch(O, data, p)
    keys = []
    for k, (lo, hi) in enumerate(bounds):
        if k % world == rank:
            b.add_part(lo, hi)
            keys.append((0, k))
    rounds = mdist.resolve_parts(b, keys)
    q.put((rank, rounds, b.fixes, [(k[1], r["cuts"].tolist(), r["entry_confirmed"]) for k, r in zip(keys, b.recs)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,name", [(2, "random"), (3, "zeros")])
def test_resolve_parts_over_gloo(world, name):
    from oracle import mi_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    data, p, bounds = _case(O, name)
    want = list(O.cdc_classic(data, p))
    parts = sorted(x for _, _, _, lst in res for x in lst)
    got = [c for _, cuts, _ in parts for c in cuts]
    assert got == want
    assert all(conf == 1 for _, _, conf in parts)
    rounds = {r for _, r, _, _ in res}
    assert len(rounds) == 1                               # every rank leaves the loop together
    fixes = sum(f for _, _, f, _ in res)
    if name == "random":
        assert rounds == {1} and fixes == 0               # every halo had re-synchronised
    else:
        assert rounds == {len(bounds)} and fixes >= len(bounds) - 1


def test_resolve_parts_local_model():
    from makisu_amd.distributed import resolve_parts_local
    from oracle import mi_oracle as O
    for name in ("random", "zeros"):
        data, p, bounds = _case(O, name)
        owners = []
        for k, (lo, hi) in enumerate(bounds):
            b = ModelBatch(O, data, p)
            b.add_part(lo, hi)
            owners.append((b, [(0, k)]))
        resolve_parts_local(owners)
        got = [c for b, _ in owners for c in b.recs[0]["cuts"].tolist()]
        assert got == list(O.cdc_classic(data, p))
