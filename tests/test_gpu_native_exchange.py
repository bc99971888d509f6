"""The library's own digest exchange (csrc/mi_comm.hip: mi_comm_init_rank / _init_all, mi_dedup_allgather
/ _allgather_all) with MORE THAN ONE RANK on a one-GPU box.

RCCL refuses two ranks on one device, so these tests load a test double in its place
(MI_RCCL_LIB=tests/rccl_stub/libmi_rccl_stub.so: the nccl* entry points over POSIX shared memory for
several processes, over device copies for several ctxs of one process).  Everything else is the
product's code: counts all-gather, padded slabs, the ragged squeeze, `first_global`, the job-wide
marking of a rank's own rows, the summed first-occurrence counts.  Checked against the oracle's
duplicate marking of the concatenated rank-major digest set.  Ragged on purpose: one rank holds many
more rows than the others (its peers' digest buffers are too small for the padded slab -> the copy
path), one rank holds NO rows, contents repeat across ranks."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "rccl_stub")
STUB = os.path.join(STUB_DIR, "libmi_rccl_stub.so")
SEED = 0x4D414B49


@pytest.fixture(scope="module")
def stub():
    src = os.path.join(STUB_DIR, "mi_rccl_stub.cpp")
    if not os.path.exists(STUB) or os.path.getmtime(STUB) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared",
                               src, "-o", STUB, "-lrt", "-lpthread"])
    return STUB


def rank_files(rank, n):
    """(sizes, content ids) of a rank: rank 0 many rows, the last rank none (n >= 3) or one tiny file,
    every rank repeats some of rank 0's contents."""
    if n >= 3 and rank == n - 1:
        return [], []
    if rank == 0:
        sizes = [65536] * 40 + [300000, 5, 0, 1 << 20]
        cids = list(range(100, 100 + len(sizes)))
        return sizes, cids
    sizes = [65536] * (3 + rank) + [2048, 300000]
    cids = [100, 101, 102] + [1000 * rank + i for i in range(rank)] + [1000 * rank + 50, 140]   # 140 = rank 0's 300000-byte file
    return sizes, cids


CHILD_COMMON = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch  # noqa: F401
import makisu_amd
from test_gpu_native_exchange import rank_files, SEED
"""

CHILD_RANK = CHILD_COMMON + r"""
rank, n, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
with makisu_amd.Engine(flags=makisu_amd.FLAG_NO_DEDUP) as e:
    if rank == 0:
        uid = e.comm_unique_id()
        open(os.path.join(d, "uid.tmp"), "wb").write(uid)
        os.rename(os.path.join(d, "uid.tmp"), os.path.join(d, "uid"))
    else:
        while not os.path.exists(os.path.join(d, "uid")):
            time.sleep(0.01)
        uid = open(os.path.join(d, "uid"), "rb").read()
    assert e.comm_ranks() == 0
    e.comm_init_rank(n, rank, uid)
    assert e.comm_ranks() == n
    sizes, cids = rank_files(rank, n)
    with e.batch() as b:
        if sizes:
            b.add_synthetic(sizes, cids, seed=SEED)
        b.run()
        for rep in range(2):                                  # twice: the exchange buffers are reused
            n_total, n_unique, first = b.dedup_allgather()
        ch = b.chunks().copy()
    np.savez(os.path.join(d, "out%%d.npz" %% rank), sha=ch["sha256"], dup=ch["dup_of"],
             scal=np.array([n_total, n_unique, first], dtype=np.int64))
    e.comm_destroy()
print("OK")
"""

CHILD_ALL = CHILD_COMMON + r"""
n, d = int(sys.argv[1]), sys.argv[2]
engines = [makisu_amd.Engine(flags=makisu_amd.FLAG_NO_DEDUP) for _ in range(n)]      # n ctxs on the one GPU
makisu_amd.comm_init_all(engines)
assert [e.comm_ranks() for e in engines] == [n] * n
batches = []
for r, e in enumerate(engines):
    sizes, cids = rank_files(r, n)
    b = e.batch()
    if sizes:
        b.add_synthetic(sizes, cids, seed=SEED)
    b.run()
    batches.append(b)
for rep in range(2):
    n_total, n_unique = makisu_amd.dedup_allgather_all(batches)
first = 0
for r, b in enumerate(batches):
    ch = b.chunks().copy()
    np.savez(os.path.join(d, "out%%d.npz" %% r), sha=ch["sha256"], dup=ch["dup_of"],
             scal=np.array([n_total, n_unique, first], dtype=np.int64))
    first += len(ch)
    b.free()
for e in engines:
    e.close()
print("OK")
"""


def _check(oracle, d, n):
    outs = [np.load(os.path.join(d, "out%d.npz" % r)) for r in range(n)]
    allrows = np.concatenate([o["sha"].reshape(-1, 32) for o in outs])
    want, want_unique = oracle.dedup_mt(allrows, 4)
    first = 0
    for r, o in enumerate(outs):
        n_total, n_unique, fg = (int(x) for x in o["scal"])
        rows = len(o["dup"])
        assert (n_total, n_unique, fg) == (len(allrows), want_unique, first), (r, n_total, n_unique, fg)
        assert np.array_equal(o["dup"], want[first:first + rows]), "rank %d: dup_of differs from the oracle" % r
        first += rows
    # the shape the test is about: ragged, a rank without rows, duplicates across ranks
    counts = [len(o["dup"]) for o in outs]
    assert max(counts) > 4 * sorted(counts)[-2] or n == 2
    assert (n < 3) or counts[-1] == 0
    assert want_unique < len(allrows)
    assert any((o["dup"] >= 0).any() and r > 0 for r, o in enumerate(outs))
    return counts


@pytest.mark.parametrize("n", [2, 3])
def test_native_exchange_n_processes_on_one_gpu(oracle, stub, tmp_path, n):
    env = dict(os.environ, MI_RCCL_LIB=stub, MI_RCCL_STUB_SLOT_MB="1")     # 1 MiB slots: rank 0's slab takes rounds
    procs = [subprocess.Popen([sys.executable, "-c", CHILD_RANK % {"root": ROOT}, str(r), str(n), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(n)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "OK" in so, so[-1000:] + se[-3000:]
    _check(oracle, str(tmp_path), n)


@pytest.mark.parametrize("n", [2, 3])
def test_native_exchange_n_ctxs_in_one_process(oracle, stub, tmp_path, n):
    """mi_comm_init_all + mi_dedup_allgather_all: the group-start / enqueue / group-end interleaving over
    n ctxs of one process -- code that had never run with n > 1."""
    env = dict(os.environ, MI_RCCL_LIB=stub)
    r = subprocess.run([sys.executable, "-c", CHILD_ALL % {"root": ROOT}, str(n), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    _check(oracle, str(tmp_path), n)


def test_stub_refuses_what_real_rccl_would_hang_on(stub, tmp_path):
    """An init_all communicator's collective outside a group cannot complete (the peers' calls come from
    the same thread): the double says so instead of pretending."""
    code = CHILD_COMMON % {"root": ROOT} + textwrap.dedent(r"""
        engines = [makisu_amd.Engine(flags=makisu_amd.FLAG_NO_DEDUP) for _ in range(2)]
        makisu_amd.comm_init_all(engines)
        b = engines[0].batch()
        b.add_synthetic([65536], [1], seed=SEED)
        b.run()
        try:
            b.dedup_allgather()                  # the single-rank call on a two-rank single-process communicator
            print("NO ERROR")
        except makisu_amd.MiError as err:
            print("REFUSED", err)
    """)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MI_RCCL_LIB=stub), capture_output=True,
                       text=True, timeout=300)
    assert "REFUSED" in r.stdout and "invalid usage" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
