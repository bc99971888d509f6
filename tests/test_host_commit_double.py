"""The HOST side of the content-aware commit -- mi_memfs_commit_layer with a ctx: walk + stage into one batch, the diff on
the batch's rows, the layer tar written from the arena (lib/snapshot/mem_fs.go:260-341 + lib/builder/step/common.go:67-111
+ lib/tario/write.go:28-52 as one flow) -- on the HIP test double, here, without a GPU (tests/hip_stub/commit_scenarios.py
says what the double can and cannot show; the roots themselves are tests/test_gpu_commit.py's business)."""
import os
import subprocess
import sys

import pytest

from test_host_hip_double import STUB_DIR, hip_double  # noqa: F401  (the fixture builds the double when stale)

WANT = ["OK scan", "OK copy", "OK many", "OK trust", "OK trust_wide", "OK slash"]


def _run(stub, tmp, threads, extra_env=None):
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + stub).strip())
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "commit_scenarios.py"), str(tmp), str(threads)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert [ln for ln in p.stdout.splitlines() if ln.startswith("OK ")] == WANT


@pytest.mark.parametrize("threads", [1, 4, 16])
def test_the_commit_reads_each_file_once_and_writes_the_references_tar(hip_double, tmp_path, threads):  # noqa: F811
    _run(hip_double, tmp_path, threads)


@pytest.mark.parametrize("env", [{"MI_WALK_INLINE": "0"}, {"MI_WALK_INLINE_MAX_KIB": "2"}, {"MI_WALK_THREADS": "1"},
                                 {"MI_WALK_INLINE_MB": "1"}, {"MI_WALK_UNSHARE": "0"}, {"MI_WALK_CLOSE_RANGE": "0"},
                                 {"MI_HIP_STUB_COPY_US": "50"}, {"MI_COMMIT_PIPELINE": "0"},
                                 {"MI_HIP_STUB_COPY_US": "200", "MI_HIP_STUB_KERNEL_US": "2000"}])
def test_whichever_way_the_bytes_reach_the_arena(hip_double, tmp_path, env):  # noqa: F811
    """the walk's knobs move files between the two ways into the arena (a directory's block / a path for the reader threads)
    and change the order in which bytes land; MI_WALK_CLOSE_RANGE=0 is the kernel without close_range (the
    directory readers then stay on the shared descriptor table); slow copies widen every window -- the tar writer then
    runs far ahead of the staging it reads from (the pipelined commit: a file's bytes are waited for); MI_COMMIT_PIPELINE=0 is
    the commit one phase after the other"""
    _run(hip_double, tmp_path, 4, env)


@pytest.mark.parametrize("trust", [False, True])
def test_a_tree_larger_than_the_device_is_scanned_in_windows(hip_double, tmp_path, trust):  # noqa: F811
    """the double refuses allocations above 6 MiB: a 9+ MB tree cannot be staged in one batch -- mi_memfs_commit_layer falls back to
    windows (MI_COMMIT_WINDOW_MB=2) for the roots and to the disk for the tar; with MI_MEMFS_TRUST_CTIME the walk that runs into the
    limit is the filtered one"""
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip(), MI_HIP_STUB_MALLOC_LIMIT_MB="6", MI_ARENA_PIECE_MB="2",
               MI_COMMIT_WINDOW_MB="2")
    if trust:
        env["MI_TEST_TRUST"] = "1"
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "commit_scenarios.py"), str(tmp_path), "4", "oversize"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK oversize" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


def test_copy_sources_larger_than_the_device_go_window_by_window(hip_double, tmp_path):  # noqa: F811
    """the double refuses allocations above 6 MiB; three COPY ops whose sources hold 7+ MB: the plan runs out of room, is made again
    without a batch, the roots come in windows (MI_COMMIT_WINDOW_MB=2) -- same tar as the header-only commit, roots in the tree"""
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip(), MI_HIP_STUB_MALLOC_LIMIT_MB="6", MI_ARENA_PIECE_MB="2",
               MI_COMMIT_WINDOW_MB="2")
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "commit_scenarios.py"), str(tmp_path), "4", "oversize_copy"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK oversize_copy" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.parametrize("variant", ["plain", "range_outgrown", "slow_mapper", "outgrown_under_a_slow_mapper"])
def test_the_arena_never_moves_while_a_walk_fills_it(hip_double, tmp_path, variant):  # noqa: F811
    """mi_arena.hip: one address range, mapped piece by piece behind the walk -- also when the range is outgrown (the pieces are
    mapped again elsewhere, with their bytes) and when the mapper is slower than the readers (a box that charges device memory by
    the byte): the layer tar holds every file's bytes"""
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip(), MI_ARENA_PIECE_MB="8")
    if variant in ("range_outgrown", "outgrown_under_a_slow_mapper"):
        env["MI_ARENA_RANGE_MB"] = "16"
    if variant in ("slow_mapper", "outgrown_under_a_slow_mapper"):       # (the second: a range that is outgrown WHILE the mapper is in the
        env["MI_HIP_STUB_MAP_US"] = "2000"                                #  middle of a piece -- the pieces move only between two of them)
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "commit_scenarios.py"), str(tmp_path), "4", "known_tree"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK known_tree" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.parametrize("case,fault", [("clean", None), ("readback1", "readback:2"), ("readback1", "readback:7"), ("readback3", "readback:3:3"),
                                        ("copy", "copy:1"), ("copy", "copy:6")])
@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_bytes_that_change_on_the_way_fail_the_commit(hip_double, tmp_path, case, fault, pipeline):  # noqa: F811
    """mi_filesum.h: a flipped byte in the read-back window is repaired by a second fetch (or fails the commit when that is wrong
    too), 4 KiB lost in HBM after the host-to-device copy fail it -- with the hop named; never a self-consistent corrupted layer
    (lib/tario/write.go:43-45: the bytes or an error)"""
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip(), MI_VERIFY_CASE=case, MI_COMMIT_PIPELINE=pipeline)
    if fault:
        env["MI_STAGE_FAULT"] = fault
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "commit_scenarios.py"), str(tmp_path), "4", "verify"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and ("OK verify " + case) in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.parametrize("n,extra", [(2, {}), (8, {}), (3, {"MI_COMMIT_PIPELINE": "0"}), (2, {"MI_STAGE_FAULT": "readback:1"}),
                                     (4, {"MI_WALK_INLINE": "0"}), (2, {"MI_COMMIT_FORCE_WINDOWS": "1", "MI_COMMIT_WINDOW_MB": "1"})])
def test_the_commit_over_several_ctxs_writes_the_one_ctx_commits_tar(hip_double, tmp_path, n, extra):  # noqa: F811
    """mi_memfs_commit_layer_n (north_star: file batches shard across the GPUs of one node; lib/snapshot/mem_fs.go:260-289): the
    walk's files spread by bytes over n ctxs, one read per file, the tar byte-identical to n = 1 and to the reference's commit; the
    handle's later commits, a COPY, windows, a repaired read-back fault -- on the HIP double, n ctxs on its one device"""
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip(), MI_TEST_N_CTXS=str(n), **extra)
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "commit_scenarios.py"), str(tmp_path), "4", "many_gpus"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and ("OK many_gpus %d" % n) in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_a_fault_at_every_place_the_injection_reaches_never_yields_a_wrong_layer(hip_double, pipeline):  # noqa: F811
    """tools/verify_fault_soak.py on the double: the k-th read-back copy flipped (once; three times running), the k-th staged span
    losing 4 KiB in HBM, k = 0..5 -- every commit is either the header-only commit's tar or MI_ERR_IO naming the hop"""
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip(), MI_COMMIT_PIPELINE=pipeline)
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "verify_fault_soak.py"), "6"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "verify fault soak: 18 commits" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
    assert "copy         failed, hop named                x 6" in p.stdout, p.stdout


@pytest.mark.parametrize("fail", ["create:0", "create:2", "map:1", "access:2", "reserve:0"])
@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_an_arena_that_cannot_be_mapped_fails_the_commit_and_nothing_hangs(hip_double, tmp_path, fail, pipeline):  # noqa: F811
    """csrc/mi_arena.hip: the mapper thread's k-th piece is refused -- everybody who waits for the mapper (reader threads, the scan's
    stage_batch, the tar writer's wait for landed bytes) gets the failure instead of waiting for ever; the pieces it did map go back.
    reserve:0 -- the process has no address range left (mi_arena.hip retires them): NOT a failure, the batch's arena is one
    allocation that moves when it grows, and the commit is the reference's"""
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip(), MI_HIP_STUB_VM_FAIL=fail,
               MI_COMMIT_PIPELINE=pipeline)
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "commit_scenarios.py"), str(tmp_path), "4", "mapper_fails"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "OK mapper_fails" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
