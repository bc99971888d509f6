"""The library's host-fed path on a HIP runtime TEST DOUBLE (tests/hip_stub/mi_hip_stub.cpp: device memory = host memory,
a stream = a thread with a queue, copies asynchronous to the caller, kernels queued no-ops), so that reader threads,
pinned slabs, the inline window, arena growth, batch reuse and two batches in flight run HERE, without a GPU -- and, with
tools/tsan_host_tests.sh, under ThreadSanitizer, where a slab refilled before its copy has completed, an arena moved
under a copy in flight or a result read before its stream was synchronised is a reported race between the stream's
thread and the caller's, not a once-in-a-thousand-runs wrong chunk count.

What is checked is what the host side answers for: every byte of every file, range and buffer lies in the batch's arena
where the file table says (mi_batch_read_back), and errors stay with their batch.  tario.WriteEntry's contract --
io.CopyN delivers exactly the file's bytes, lib/tario/write.go:43-45."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "hip_stub")
STUB = os.path.join(STUB_DIR, "libmi_hip_stub.so")


@pytest.fixture(scope="module")
def hip_double(engine_lib):
    src = os.path.join(STUB_DIR, "mi_hip_stub.cpp")
    if not os.path.exists(STUB) or os.path.getmtime(STUB) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__",
                               "-I/opt/rocm/include", src, "-o", STUB, "-lpthread"])
    return STUB


def _run(stub, tmp, threads, slab, extra_env=None):
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + stub).strip())
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "scenarios.py"), str(tmp), str(threads), str(slab)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    assert [ln for ln in p.stdout.splitlines() if ln.startswith("OK ")] == ["OK mix", "OK growth", "OK two", "OK interleaved", "OK errors", "OK api", "OK warm", "OK tree", "OK two_ctxs"]


@pytest.mark.parametrize("threads,slab", [(1, 65536), (4, 65536), (16, 131072), (3, 1 << 20)])
def test_every_staged_byte_lands_where_the_file_table_says(hip_double, tmp_path, threads, slab):
    _run(hip_double, tmp_path, threads, slab)


def test_with_the_walk_handing_every_file_over_as_a_path(hip_double, tmp_path):
    """MI_WALK_INLINE=0: no file is read where it is listed (round 3's way); the same bytes, the same order"""
    _run(hip_double, tmp_path, 4, 65536, {"MI_WALK_INLINE": "0"})


def test_with_the_shared_descriptor_table(hip_double, tmp_path):
    """MI_WALK_UNSHARE=0: the directory readers keep the process's table -- lstat first, then open, one file at a time"""
    _run(hip_double, tmp_path, 4, 65536, {"MI_WALK_UNSHARE": "0"})


def test_with_a_descriptor_limit_below_the_directory_sizes(hip_double, tmp_path):
    """a soft RLIMIT_NOFILE of 128: the 200-file directories of the scenario do not fit a reader's table -- they are read
    the lstat-first way, one descriptor at a time"""
    import resource
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip())
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "scenarios.py"), str(tmp_path), "4", "65536"], env=env,
                       capture_output=True, text=True, timeout=900,
                       preexec_fn=lambda: resource.setrlimit(resource.RLIMIT_NOFILE, (128, hard)))
    assert p.returncode == 0 and "OK tree" in p.stdout, p.stdout[-800:] + p.stderr[-1500:]


def test_an_adder_that_outruns_the_readers_waits_for_descriptors(hip_double, tmp_path):
    """mi_batch_add_path opens the file in the call and the descriptor travels with the file's pieces until they are read: with
    a table of 40 descriptors and copies slowed down, 300 adds in a row hold more descriptors than there are -- the call waits
    for the readers instead of failing with EMFILE (a full table that is the caller's own queue is not the caller's error)"""
    import resource
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip(), MI_HIP_STUB_COPY_US="1500")
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "scenarios.py"), str(tmp_path), "2", "65536", "interleaved"], env=env,
                       capture_output=True, text=True, timeout=900,
                       preexec_fn=lambda: resource.setrlimit(resource.RLIMIT_NOFILE, (40, hard)))
    assert p.returncode == 0 and "OK interleaved" in p.stdout, p.stdout[-800:] + p.stderr[-1500:]


def test_with_a_block_budget_that_runs_out(hip_double, tmp_path):
    """MI_WALK_INLINE_MB=1: after a megabyte of blocks alive the directories' files go as paths again -- mixed ways"""
    _run(hip_double, tmp_path, 4, 65536, {"MI_WALK_INLINE_MB": "1"})


def test_a_block_goes_back_to_the_pool_when_it_is_copied_not_when_the_walk_ends(hip_double, tmp_path):
    """A directory's block is let go of when the reader threads have copied it: a walk over 100 directories cycles through
    a few tens of blocks.  (Until round 4's last session the walk's own record of a directory kept its block until the
    walk ended: every directory needed a block of its own -- a process's first walk touched as much fresh block memory as
    the tree holds small files, and MI_WALK_INLINE_MB bounded the small files of a whole TREE that could take this way,
    not the blocks alive at a time.)"""
    import re
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip(),
               MI_WALK_THREADS="4", MI_WALK_TIMING="1", MI_ARENA_PIECE_MB="2")    # (small pieces: the double's hipMemCreate fills a
                                                                                  #  piece byte by byte -- readers that wait for a 32 MiB
                                                                                  #  fill on a busy machine let blocks pile up behind them)
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "scenarios.py"), str(tmp_path), "4", str(1 << 20), "recycle"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "OK recycle" in p.stdout, p.stdout[-800:] + p.stderr[-1500:]
    line = [ln for ln in p.stderr.splitlines() if ln.startswith("mi_walk:")][0]
    m = re.search(r"(\d+) blocks carved, .* at most ([0-9.]+) MB of blocks alive, (\d+) files read into blocks", line)
    assert m, line
    assert int(m.group(3)) == 10000, line                   # every file went the block way (120 MB, far below the budget)
    assert int(m.group(1)) <= 60, line                      # ... through far fewer blocks than there are directories
    assert float(m.group(2)) <= 0.6 * 120, line


def test_the_block_pool_gives_its_memory_back_between_walks(hip_double, tmp_path):
    """the pool of the walk's blocks belongs to the process and outlives every walk -- a long-running host must
    not keep half a gigabyte resident for ever after its first one.  Between walks at most MI_WALK_POOL_IDLE_MB (default 64)
    stay; the rest of the pages go back (the carved blocks keep their address range)."""
    import re
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip(),
               MI_WALK_THREADS="4", MI_WALK_TIMING="1", MI_WALK_POOL_IDLE_MB="2")
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "scenarios.py"), str(tmp_path), "4", str(1 << 20), "recycle"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "OK recycle" in p.stdout, p.stdout[-800:] + p.stderr[-1500:]
    lines = [ln for ln in p.stderr.splitlines() if ln.startswith("mi_walk: block pool between walks")]
    assert lines, p.stderr[-1500:]
    for ln in lines:
        m = re.search(r"([0-9.]+) MB resident \(limit 2\)", ln)
        assert m and float(m.group(1)) <= 2.2, ln


def test_with_slow_copies(hip_double, tmp_path):
    """every queued copy takes 100 us longer: what returns early shows"""
    _run(hip_double, tmp_path, 8, 65536, {"MI_HIP_STUB_COPY_US": "100"})


def test_the_double_is_not_the_product():
    """nothing under makisu_amd/ or include/ knows about the double"""
    for base in ("makisu_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".h")):
                    assert "hip_stub" not in open(os.path.join(dp, fn), errors="replace").read(), fn


@pytest.mark.parametrize("threads,slab", [(4, 1 << 20), (2, 65536)])
def test_strings_too_long_for_a_gpu_lane_are_hashed_by_host_streams(hip_double, tmp_path, threads, slab):
    """MI_FLAG_FILE_SHA256 and mi_sha256_many route long strings to SHA-NI streams (the reader threads out of HBM;
    host threads out of the caller's memory) -- their digests are hashlib's, on the double too"""
    env = dict(os.environ, LD_PRELOAD=(os.environ.get("LD_PRELOAD", "") + " " + hip_double).strip())
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "scenarios.py"), str(tmp_path), str(threads), str(slab), "long_strings"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "OK long_strings" in p.stdout, p.stdout[-800:] + p.stderr[-1500:]
