"""The library's own digest exchange (csrc/mi_comm.hip: mi_comm_init_rank / _init_all, mi_dedup_allgather
/ _allgather_all) with MORE THAN ONE RANK on a one-GPU box.

RCCL refuses two ranks on one device, so these tests load a test double in its place
(MI_RCCL_LIB=tests/rccl_stub/libmi_rccl_stub.so: the nccl* entry points over POSIX shared memory for
several processes, over device copies for several ctxs of one process).  Everything else is the
product's code: counts all-gather, padded slabs, the ragged squeeze, `first_global`, the job-wide
marking of a rank's own rows, the summed first-occurrence counts.  Checked against the oracle's
duplicate marking of the concatenated rank-major digest set.  Ragged on purpose: one rank holds many
more rows than the others (its peers' digest buffers are too small for the padded slab -> the copy
path), one rank holds NO rows, contents repeat across ranks.

Round 4: the rank count BASELINE.json's configs[3] / [4] name -- EIGHT -- in both forms
(8 processes; 8 ctxs of one process): the ragged shape with one rank holding 8x the rows of the others and one
holding none, C4's eight `index mod 8` shards (20 000 files each), and the Zipf C5 mix with its large files
split into 8 parts (`split_threshold` 32 MiB; the owners agree on the boundary cuts over a gloo group or, in
one process, through resolve_parts_local) -- `dup_of`, `first_global`, `n_total`, `n_unique` against the
oracle's marking of the rank-major concatenation; and `bench.py --gpus 8` launched BARE (no torchrun) prints a
C4 line of the library's own exchange with `rccl_ranks: 8`."""
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
    """(sizes, content ids) of a rank: rank 0 many rows (at 8 ranks: 8x the rows of the next largest), the
    last rank none (n >= 3) or one tiny file, every rank repeats some of rank 0's contents."""
    if n >= 3 and rank == n - 1:
        return [], []
    if rank == 0:
        sizes = [65536] * (40 if n < 8 else 160) + [300000, 5, 0, 1 << 20]
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

CHILD_COMMON += r"""
from makisu_amd import workloads as W
from makisu_amd import distributed as mdist

def shard_of(case, rank, n):
    if case == "c4":
        return W.c4(rank, n, 20000)
    return W.c5(rank, n, 1 * W.GIB, split_threshold=32 * W.MIB)          # 8 GiB job-wide at n = 8

def fill(case, b, rank, n):
    # adds the rank's items; returns the keys of its parts ([] without split files)
    if case == "ragged":
        sizes, cids = rank_files(rank, n)
        if sizes:
            b.add_synthetic(sizes, cids, seed=SEED)
        return []
    return W.fill_batch(b, shard_of(case, rank, n))
"""

CHILD_RANK = CHILD_COMMON + r"""
rank, n, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
case = sys.argv[4] if len(sys.argv) > 4 else "ragged"
if case == "c5":                                              # the parts' owners talk over a host group
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + os.path.join(d, "store"), rank=rank, world_size=n)
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
    with e.batch() as b:
        keys = fill(case, b, rank, n)
        if case == "c5":
            mdist.resolve_parts(b, keys)
        b.run()
        form = os.environ.get("MI_TEST_EXCHANGE_FORM", "allgather")
        if form == "alltoall":                                # the all-gather form first: the other must leave the same column
            want = b.dedup_allgather()
            want_dup = b.chunks()["dup_of"].copy()
        for rep in range(2):                                  # twice: the exchange buffers are reused
            n_total, n_unique, first = b.dedup_alltoall() if form == "alltoall" else b.dedup_allgather()
        if form == "alltoall":
            assert (n_total, n_unique, first) == want, ((n_total, n_unique, first), want)
            assert np.array_equal(b.chunks()["dup_of"], want_dup)
        ga, ma = e.comm_exchange_ms()
        assert ga >= 0 and ma >= 0 and (ga + ma > 0 or n_total == 0), (ga, ma)
        ch = b.chunks().copy()
    np.savez(os.path.join(d, "out%%d.npz" %% rank), sha=ch["sha256"], dup=ch["dup_of"],
             scal=np.array([n_total, n_unique, first], dtype=np.int64))
    e.comm_destroy()
print("OK")
"""

CHILD_ALL = CHILD_COMMON + r"""
n, d = int(sys.argv[1]), sys.argv[2]
case = sys.argv[3] if len(sys.argv) > 3 else "ragged"
engines = [makisu_amd.Engine(flags=makisu_amd.FLAG_NO_DEDUP) for _ in range(n)]      # n ctxs on the one GPU
makisu_amd.comm_init_all(engines)
assert [e.comm_ranks() for e in engines] == [n] * n
batches, owners = [], []
for r, e in enumerate(engines):
    b = e.batch()
    owners.append((b, fill(case, b, r, n)))
    batches.append(b)
if case == "c5":
    assert sum(len(k) for _, k in owners) >= n                # at least one file really is split into n parts
    mdist.resolve_parts_local(owners)
for b in batches:
    b.run()
form = os.environ.get("MI_TEST_EXCHANGE_FORM", "allgather")
if form == "alltoall":
    want = makisu_amd.dedup_allgather_all(batches)
    want_dup = [b.chunks()["dup_of"].copy() for b in batches]
for rep in range(2):
    n_total, n_unique = makisu_amd.dedup_allgather_all(batches, form=form)
if form == "alltoall":
    assert (n_total, n_unique) == want, ((n_total, n_unique), want)
    assert all(np.array_equal(b.chunks()["dup_of"], w) for b, w in zip(batches, want_dup))
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


def _check(oracle, d, n, case="ragged"):
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
    counts = [len(o["dup"]) for o in outs]
    if case == "ragged":
        # the shape the test is about: ragged, a rank without rows, duplicates across ranks
        assert max(counts) > 4 * sorted(counts)[-2] or n == 2
        assert n < 8 or max(counts) >= 8 * sorted(counts)[-2]       # the padded slab dwarfs every peer's digest buffer
        assert (n < 3) or counts[-1] == 0
        assert want_unique < len(allrows)
        assert any((o["dup"] >= 0).any() and r > 0 for r, o in enumerate(outs))
    elif case == "c4":
        assert min(counts) > 100000 and len(allrows) - want_unique <= 2     # distinct contents (1-byte tails may coincide)
    else:
        assert want_unique < len(allrows) // 2                       # 90 % of the files are copies
        assert any((o["dup"] >= 0).any() and r > 0 for r, o in enumerate(outs))
    return counts


CASES = [(2, "ragged"), (3, "ragged"), (8, "ragged"), (8, "c4"), (8, "c5")]
FORMS = ["allgather", "alltoall"]          # mi_dedup_allgather[_all]; mi_dedup_alltoall[_all] -- the hash-partitioned form, held
                                           # against the oracle AND, in the same process, against the all-gather form's column


@pytest.mark.timeout(900)
@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("n,case", CASES)
def test_native_exchange_n_processes_on_one_gpu(oracle, stub, tmp_path, n, case, form):
    env = dict(os.environ, MI_RCCL_LIB=stub, MI_RCCL_STUB_SLOT_MB="1", MI_TEST_EXCHANGE_FORM=form)   # 1 MiB slots: rank 0's slab / share takes rounds
    procs = [subprocess.Popen([sys.executable, "-c", CHILD_RANK % {"root": ROOT}, str(r), str(n), str(tmp_path), case],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(n)]
    outs = [p.communicate(timeout=800) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "OK" in so, so[-1000:] + se[-3000:]
    _check(oracle, str(tmp_path), n, case)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("n,case", CASES)
def test_native_exchange_n_ctxs_in_one_process(oracle, stub, tmp_path, n, case, form):
    """mi_comm_init_all + mi_dedup_allgather_all over n ctxs of one process: host-known counts, every
    allocation and pad copy before the group, the group holding the n all-gathers and nothing else.  (form "alltoall":
    mi_dedup_alltoall_all -- the shares counted on the devices, two groups of sends and receives.)"""
    env = dict(os.environ, MI_RCCL_LIB=stub, MI_TEST_EXCHANGE_FORM=form)
    r = subprocess.run([sys.executable, "-c", CHILD_ALL % {"root": ROOT}, str(n), str(tmp_path), case], env=env,
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    _check(oracle, str(tmp_path), n, case)


@pytest.mark.parametrize("form", FORMS)
def test_plain_c_exchange(oracle, stub, tmp_path, form):
    """The single-process form from plain C (tests/cabi/exchange_driver.c: one ctx per rank, a host thread per rank for
    the scan, mi_comm_init_all, mi_dedup_allgather_all twice) with 8 ranks on this GPU: the printed dup_of of every rank
    is the oracle's marking of the rank-major concatenation; half of the job's chunks repeat the previous rank's."""
    exe = str(tmp_path / "exchange_driver")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi", "exchange_driver.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "makisu_amd"), "-lmakisu_mi", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "makisu_amd")])
    n, per = 8, 600
    out = subprocess.run([exe, str(n), str(per), "0"], env=dict(os.environ, MI_RCCL_LIB=stub, MI_EXCHANGE_FORM=form),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-2000:]
    lines = out.stdout.splitlines()
    ranks = [ln.split() for ln in lines if ln.startswith("K ")]
    rows = [ln.split() for ln in lines if ln.startswith("C ")]
    n_total, n_unique, seen = (int(x) for x in lines[-1].split()[1:])
    assert lines[-1].startswith("T ") and seen == n and len(ranks) == n and len(rows) == n_total
    sha = np.array([bytearray.fromhex(r[3][len("sha256:"):]) for r in rows], dtype=np.uint8)
    want, want_unique = oracle.dedup_mt(sha, 4)
    assert np.array_equal(np.array([int(r[2]) for r in rows], dtype=np.int64), want) and n_unique == want_unique
    first = 0
    for r, (_, rank, nc, fg, dups, gather, marking) in enumerate(ranks):
        assert (int(rank), int(fg)) == (r, first) and float(marking) > 0
        assert r == 0 or int(dups) > int(nc) // 3                    # the first half of the rank repeats its predecessor
        first += int(nc)
    assert n_unique < 0.65 * n_total


def _bench_line(out):
    import json
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.timeout(900)
def test_bench_bare_gpus_8_is_the_library_exchange(stub):
    """`python bench.py --gpus 8` WITHOUT torchrun: one process, eight ctxs (all on this GPU here), the library's
    own communicator (mi_comm_init_all) and exchange -- a C4 line with rccl_ranks 8, the closed-form unique
    count, per-rank step / exchange / marking times and efficiency_vs_n1 from the same run."""
    env = dict(os.environ, MI_BENCH_FORCE_DEVICE="0", MI_RCCL_LIB=stub)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--files", "20000",
                          "--steps", "2", "--warmup", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    j = _bench_line(out)
    assert j["n_gpus"] == 8 and j["config"]["name"] == "c4" and j["config"]["rccl_ranks"] == 8
    assert j["config"]["exchange"] == "native" and "single process" in j["config"]["launch"]
    assert j["config"]["exchange_form"] == "allgather"                      # the default form
    assert j["config"]["job_bytes_per_step"] == 8 * 20000 * 65536
    assert j["dedup_check"]["ok"] and j["dedup_check"]["n_total"] == sum(j["config"]["chunks_per_rank_last_batch"])
    for k in ("step_ms", "exchange_gather_ms", "marking_ms", "sha_chunks_ms"):       # (the double gathers on the host: ~0 ms)
        assert len(j["per_rank"][k]) == 8 and all(v > 0 or k == "exchange_gather_ms" for v in j["per_rank"][k]), (k, j["per_rank"][k])
    assert j["n1_same_run"]["value"] > 0 and 0 < j["efficiency_vs_n1"] < 1.5
    assert j["roofline"]["frac"] > 0
    assert j["config"]["rccl_ranks_per_ctx"] == [8] * 8                     # counted (ncclCommCount of every ctx), not assumed
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] >= 1   # the host's scanner in the same run, at N > 1 too
    assert "launch_note" not in j["config"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("launch", ["bare", "torchrun"])
def test_bench_with_the_all_to_all_form(stub, launch):
    """`bench.py --exchange-form alltoall`, bare (4 ctxs of one process: mi_dedup_alltoall_all) and as the driver launches it
    (one process per rank: mi_dedup_alltoall): the job-wide unique count is the generator's closed form, the line says which
    form ran, and the per-rank wire / device-work times are there."""
    env = dict(os.environ, MI_BENCH_FORCE_DEVICE="0", MI_RCCL_LIB=stub, MASTER_ADDR="127.0.0.1")
    tail = ["--gpus", "4", "--files", "8000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--exchange-form", "alltoall"]
    if launch == "bare":
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
               "--master-port", "29591", os.path.join(ROOT, "bench.py")] + tail
    j = _bench_line(subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800))
    assert j["n_gpus"] == 4 and j["config"]["exchange"] == "native" and j["config"]["exchange_form"] == "alltoall"
    assert j["config"]["rccl_ranks"] == 4 and j["dedup_check"]["ok"], j["dedup_check"]
    assert len(j["per_rank"]["marking_ms"]) == 4 and all(v > 0 for v in j["per_rank"]["marking_ms"])
    assert "launch_note" not in j["config"] and "exchange_note" not in j["config"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fault,says", [("fail", "mi_comm_init_all over 4 devices failed"),
                                        ("hang", "mi_comm_init_all over 4 devices did not return within"),
                                        ("exchange-hangs", "the first mi_dedup_allgather_all over 4 ranks did not return within")])
def test_bench_bare_first_contact_always_ends_in_a_line(stub, fault, says):
    """First contact with N GPUs may not work in the single-process form (ncclCommInitAll failing, or hanging, or the first
    grouped collective never completing): the bare job says what happened and re-executes itself under
    torch.distributed.run, whose ranks bring the communicator up one by one -- a line comes out, and it says which path
    produced it.  (The double plays the faulty library: MI_RCCL_STUB_INIT_ALL.)"""
    env = dict(os.environ, MI_BENCH_FORCE_DEVICE="0", MI_RCCL_LIB=stub, MI_RCCL_STUB_INIT_ALL=fault)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--files", "8000", "--steps", "2",
                          "--warmup", "1", "--watchdog-s", "8", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=800)
    j = _bench_line(out)
    assert j["n_gpus"] == 4 and j["config"]["rccl_ranks"] == 4 and j["config"]["launch"] == "one process per GPU"
    assert says in j["config"]["launch_note"] and "re-executed by the bare single-process form" in j["config"]["launch_note"]
    assert j["dedup_check"]["ok"] and "re-executing under torch.distributed.run" in out.stderr


@pytest.mark.timeout(900)
def test_bench_bare_refuses_without_enough_ranks(stub):
    """No line when the node cannot give the job its N devices (here: one GPU, no forced device)."""
    env = dict(os.environ, MI_RCCL_LIB=stub)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MI_BENCH_FORCE_DEVICE"):
        env.pop(k, None)
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this node has 8 devices")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--files", "2000"], env=env,
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "device(s)" in out.stderr


@pytest.mark.timeout(900)
def test_bench_torchrun_defaults_to_the_native_exchange(stub):
    """The driver's launch line (torch.distributed.run, one process per GPU): the exchange is the library's
    (mi_comm_init_rank + mi_dedup_allgather), torch ships the id and runs the host-side barrier over gloo."""
    env = dict(os.environ, MI_BENCH_FORCE_DEVICE="0", MI_RCCL_LIB=stub, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
           "--master-addr", "127.0.0.1", "--master-port", "29583", os.path.join(ROOT, "bench.py"),
           "--gpus", "4", "--files", "20000", "--steps", "2", "--warmup", "1"]
    j = _bench_line(subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800))
    assert j["n_gpus"] == 4 and j["config"]["name"] == "c4" and j["config"]["rccl_ranks"] == 4
    assert j["config"]["exchange"] == "native" and "gloo" in j["config"]["exchange_backend"]
    assert "exchange_note" not in j["config"]
    assert j["dedup_check"]["ok"], j["dedup_check"]
    assert len(j["per_rank"]["step_ms"]) == 4 and all(v > 0 for v in j["per_rank"]["marking_ms"])
    assert len(j["n1_same_run"]["per_rank_value"]) == 4 and j["efficiency_vs_n1"] > 0


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


@pytest.mark.timeout(900)
def test_bench_torchrun_c5_split_files_through_the_native_exchange(stub):
    """bench.py --gpus 2 --config c5 as the driver would launch it: the Zipf mix LPT-sharded over two ranks, files >= 32 MiB
    split into two parts whose owners agree on the boundary cuts (resolve_parts over the gloo plumbing), and the digest
    exchange the LIBRARY's (mi_dedup_allgather on the double) -- the job-wide unique count = the generator's closed form."""
    env = dict(os.environ, MI_BENCH_FORCE_DEVICE="0", MI_RCCL_LIB=stub, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29585", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--config", "c5", "--bytes-per-gpu", "3", "--split-mib", "32", "--steps", "2", "--warmup", "1"]
    j = _bench_line(subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800))
    assert j["n_gpus"] == 2 and j["config"]["name"] == "c5" and j["config"]["exchange"] == "native" and j["config"]["rccl_ranks"] == 2
    assert j["config"]["parts_this_rank"] > 0 and j["config"]["files_split_into_parts_job"] > 0
    assert j["dedup_check"]["ok"], j["dedup_check"]
    assert j["n1_same_run"]["value"] > 0 and len(j["per_rank"]["marking_ms"]) == 2


@pytest.mark.timeout(900)
def test_bench_falls_back_to_the_torch_driver_when_a_rank_loses_its_communicator(stub):
    """A first contact with several GPUs must yield a line: if the library's communicator does not come up on SOME rank, every
    rank hears of it (over the host group), drops its communicator and the torch.distributed driver runs the same exchange
    -- the line says so.  Simulated on rank 1 of 2 (both on this GPU; the torch driver over gloo here)."""
    env = dict(os.environ, MI_BENCH_FORCE_DEVICE="0", MI_RCCL_LIB=stub, MASTER_ADDR="127.0.0.1", MI_BENCH_FAIL_NATIVE_ON_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29587", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--files", "20000", "--steps", "2", "--warmup", "1", "--backend", "gloo"]
    j = _bench_line(subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800))
    assert j["config"]["exchange"] == "torch" and "simulated failure" in j["config"]["exchange_note"]
    assert j["config"]["rccl_ranks"] is None                      # gloo counted the ranks, not RCCL: the line cannot claim it
    assert j["dedup_check"]["ok"] and j["n_gpus"] == 2
    # ... and when rank 0 cannot even create the communicator id (no collective library to load), its peers, who wait for
    # the id, are told so instead of waiting for ever
    env = dict(env, MI_RCCL_LIB="/nonexistent/librccl.so")
    env.pop("MI_BENCH_FAIL_NATIVE_ON_RANK")
    cmd[cmd.index("29587")] = "29589"
    j = _bench_line(subprocess.run(cmd + ["--no-n1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300))
    assert j["config"]["exchange"] == "torch" and "could not create the communicator id" in j["config"]["exchange_note"]
    assert j["dedup_check"]["ok"]
