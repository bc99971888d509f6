#!/usr/bin/env python3
"""bench.py -- GiB/s hashed (Gear CDC + SHA-256 per chunk) on MI355X, BASELINE.json's metric.

One "step" = one pass of the hot path over one rank's share of the config, already resident in
HBM: Gear marking + cut selection -> chunk table -> SHA-256 per chunk -> per-file roots ->
duplicate marking (+ for N > 1 the all-gather of the chunk-digest set over RCCL and the job-wide
duplicate marking).

  --config c2   BASELINE.json configs[1]: 100 000 x 64 KiB per GPU          (default at N = 1:
                the configuration the metric is quoted on)
  --config c3   configs[2]: 1 000 x 128 MiB per GPU, as two half-batches in flight; also reports
                the HOST-FED rate (page-cache files -> pinned staging -> H2D overlapped with the
                scan) next to the resident one
  --config c4   configs[3]: N x 1.25 M x 64 KiB, file index mod N            (default at N > 1)
  --config c5   configs[4]: sizes Zipf(s = 1.1) over the log2 buckets 2^10..2^30 B, 90 % of the files
                copies of the other 10 %, LPT shards (with N > 1 files >= 256 MiB are split into one
                part per GPU); the job-wide unique-chunk count is checked against the generator's
                closed form.  --config c5u: rounds 1-2's log-uniform stand-in 2^U(10,30)

How it is launched decides who runs the N > 1 job (the digest exchange is the LIBRARY's own RCCL binding,
csrc/mi_comm.hip, either way -- what a Go host calls; `--exchange torch` keeps the torch.distributed driver):
  python bench.py --gpus N                       ONE process, N ctxs, one host thread per device:
                                                 mi_comm_init_all + mi_dedup_allgather_all, no torch in the job
  python -m torch.distributed.run ... bench.py   one process per GPU: mi_comm_init_rank + mi_dedup_allgather; torch
                                                 ships the 128-byte id and runs the host-side barrier (gloo)
Both refuse to print a line unless ncclCommCount says N, and both first run the N = 1 form of the same per-GPU work
(`n1_same_run`) so that the line carries `efficiency_vs_n1`.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (sha256_items_kernel, chunk pass): algorithmic bytes
                  per launch / its average launch duration measured with HIP events on the
                  engine's own stream, against the 8 TB/s HBM3E peak; plus the VALU roof that
                  actually binds SHA-256.
  cpu_baseline -- the CPU oracle (a port; the Go reference cannot be built here) doing the
                  same work on a bounded sample on this node's host cores.
Only the cpu_baseline leg touches oracle/.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one process per GPU shares device memory with its peers (RCCL over xGMI) through dmabuf handles: the host driver of this
# pool supports nothing else, and the HIP runtime reads the switch when it starts -- so before anything loads it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SEED = 0x4D414B49
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
# The VALU roof of SHA-256 (the 64-round compression alone, 8 waves/SIMD, no memory traffic) is
# MEASURED IN THIS RUN on this device (mi_sha_valu_roof, ~30 ms, right after the timed region);
# round 1's figure from tools/ubench_sha.hip on another box was 1767 GB/s.


class ClockSampler:
    """sclk / socket power of one GPU from sysfs (hwmon freq1_input, power1_average | power1_input),
    sampled every 5 ms by a thread while a region runs; rocm-smi once if sysfs has nothing."""

    def __init__(self, pci_bus_id=None):
        import glob
        self.freq, self.power = None, None
        for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(card + "/vendor").read().strip() != "0x1002":
                    continue
                real = os.path.realpath(card)
                if pci_bus_id and os.path.basename(real).lower() != pci_bus_id.lower():
                    continue
                for hw in glob.glob(card + "/hwmon/hwmon*"):
                    f = hw + "/freq1_input"
                    pw = [x for x in (hw + "/power1_average", hw + "/power1_input") if os.path.exists(x)]
                    if os.path.exists(f):
                        self.freq, self.power = f, (pw[0] if pw else None)
                        break
                if self.freq:
                    break
            except OSError:
                continue
        # memory-side clocks: pp_dpm_{mclk,fclk,socclk} list the levels, the current one starred
        self.dpm = {}
        if self.freq:
            dev = os.path.dirname(os.path.dirname(os.path.dirname(self.freq)))
            for k in ("mclk", "fclk", "socclk"):
                f = os.path.join(dev, "pp_dpm_" + k)
                if os.path.exists(f):
                    self.dpm[k] = f
        self.dpm_samples = []
        self.samples = []
        self._stop = None

    def _read_dpm(self):
        import re
        out = {}
        for k, f in self.dpm.items():
            try:
                m = re.search(r"(\d+)\s*mhz\s*\*", open(f).read(), re.I)
                if m:
                    out[k] = int(m.group(1))
            except OSError:
                pass
        return out

    def _read(self):
        try:
            mhz = int(open(self.freq).read()) / 1e6 if self.freq else None
            w = int(open(self.power).read()) / 1e6 if self.power else None
            return mhz, w
        except (OSError, ValueError):
            return None, None

    def start(self):
        import threading
        self.samples = []
        self.dpm_samples = []
        if not self.freq:
            return
        self._stop = threading.Event()

        def loop():
            i = 0
            while not self._stop.is_set():
                self.samples.append(self._read())
                if self.dpm and i % 8 == 0:                     # the SMU answers these; every 40 ms is enough
                    self.dpm_samples.append(self._read_dpm())
                i += 1
                self._stop.wait(0.005)
        self._t = threading.Thread(target=loop, daemon=True)
        self._t.start()

    def stop(self):
        if self._stop is None:
            return self.smi_once()
        self._stop.set()
        self._t.join()
        self._stop = None
        mhz = [a for a, _ in self.samples if a]
        w = [b for _, b in self.samples if b]
        out = {"source": "sysfs hwmon (freq1_input, %s), %d samples at 5 ms" %
                         (os.path.basename(self.power) if self.power else "no power file", len(self.samples))}
        if mhz:
            out.update({"sclk_mhz_min": round(min(mhz)), "sclk_mhz_mean": round(sum(mhz) / len(mhz)), "sclk_mhz_max": round(max(mhz))})
        if w:
            out.update({"power_w_mean": round(sum(w) / len(w)), "power_w_max": round(max(w))})
        for k in self.dpm:
            v = [d[k] for d in self.dpm_samples if k in d]
            if v:
                out[k + "_mhz_min_max"] = [min(v), max(v)]
        return out

    @staticmethod
    def smi_once():
        import re
        import subprocess
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True,
                                 timeout=20).stdout
        except Exception as e:                                  # noqa: BLE001
            return {"source": "unavailable (%s)" % type(e).__name__}
        out = {"source": "rocm-smi --showclocks --showpower, one sample AFTER the region (no sysfs hwmon here)"}
        m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
        if m:
            out["sclk_mhz_after"] = int(m.group(1))
        m = re.search(r"Power \(W\): ([0-9.]+)", txt)
        if m:
            out["power_w_after"] = float(m.group(1))
        return out


def default_inflight(config, exchange):
    """Batches in flight when --inflight is not given: TWO -- the steady state of a host that has the next layer
    ready while one is scanned (the second batch's passes fill the first one's tails, and an exchange hides behind
    the other batch's kernels); three for the small config with an exchange.  `value` is that
    rate, as in rounds 1-2.  With two batches in flight their persistent hashing grids share the SIMDs, so the span
    between a launch's events (6-8 ms) says nothing about the kernel (4.2 ms): `roofline` therefore comes from
    the SERIAL steps run right after the timed region (one batch at a time, the kernel has the GPU to itself),
    and `one_batch_at_a_time` carries round 3's form of the headline."""
    if config == "c3" and not exchange:
        return 1            # c3's files are ONE step: in flight means halves; the one 134 GB batch is the faster form (1 116 vs 1 106 GiB/s)
    if not exchange:
        return 2
    return 3 if config == "c2" else 2


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by a cgroup CPU quota
    (cpu.max / cpu.cfs_quota_us) when one is set -- os.cpu_count() alone reports the node's cores
    even inside a container that is throttled to a few of them."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt and txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except (OSError, ValueError, IndexError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(shard, budget_files=None):
    """The oracle (a port of the same spec) on a bounded sample of the SAME workload, timed around
    the C call only.  Three numbers: all host cores (one file per thread at a time, threaded
    bucketed duplicate marking), one thread, and the reference-shaped single stream (what Makisu
    does today: one running SHA-256 over the tar-framed files, lib/builder/step/common.go:35-63)."""
    from oracle import mi_oracle as O
    O.build()
    cores, quota = usable_cores()
    p = O.CdcParams(SEED, 13, 2048, 65536)
    # sample: a prefix of this rank's files, at most ~6.5 GB of host memory (all of C2)
    sizes = shard.sizes
    keep = np.cumsum(sizes) <= 6_600_000_000
    n = max(1, int(keep.sum()))
    if budget_files:
        n = min(n, budget_files)
    sizes, cids = sizes[:n], shard.cids[:n]
    t0 = time.perf_counter()
    data, offs = O.synth_fill_many(shard.seed, cids, sizes, cores)
    gen_s = time.perf_counter() - t0
    nbytes = int(sizes.sum())
    best, phases, n_chunks = None, None, 0
    for _ in range(3):
        t0 = time.perf_counter()
        _, chunks = O.scan_batch(data, offs, sizes, p, True, cores, 0)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, phases, n_chunks = dt, O.last_phase_seconds(), len(chunks)
    # one thread: a slice of the sample that takes about a second
    n1 = max(1, min(n, int(1.0e9 // max(1, nbytes // n))))
    b1 = int(sizes[:n1].sum())
    t0 = time.perf_counter()
    O.scan_batch(data[:b1], offs[:n1], sizes[:n1], p, True, 1, 0)
    rate1 = b1 / (time.perf_counter() - t0)
    # reference-shaped: ONE stream over the whole sample (bounded to ~4 GB: it is one core)
    nr = max(1, int((np.cumsum(sizes) <= 4_000_000_000).sum()))
    br = int(sizes[:nr].sum())
    t0 = time.perf_counter()
    O.layer_scan(data[:br], offs[:nr], sizes[:nr], True)
    ref_rate = br / (time.perf_counter() - t0)
    scan_rate = nbytes / max(phases["scan_s"], 1e-9)
    return {"value": round(nbytes / best / 2**30, 3), "unit": "GiB/s", "cores": cores,
            "kind": "port",
            "sample": "first %d files of this rank's %s shard (%.0f MiB): Gear CDC + SHA-256 per chunk + "
                      "roots on all cores (one file per thread at a time), duplicate marking in 4096 "
                      "digest-prefix buckets across threads; best of 3, timed around the C call, data "
                      "already in host memory (generated in %.1f s, untimed)"
                      % (n, shard.name, nbytes / 2**20, gen_s),
            "sha_ni": bool(O.have_shani()), "node_cores": os.cpu_count(), "cgroup_cpu_quota": quota,
            "phase_s": {k: round(v, 4) for k, v in phases.items()},
            "scan_phase_GiBps": round(scan_rate / 2**30, 3),
            "single_thread_GiBps": round(rate1 / 2**30, 3),
            "parallel_efficiency": round(nbytes / best / (rate1 * cores), 3),
            "reference_shaped_single_stream_GiBps": round(ref_rate / 2**30, 3),
            "reference_shaped_sample_MiB": round(br / 2**20),
            "n_chunks_sample": int(n_chunks)}


def host_fed_rate(eng, n_files=48, file_bytes=128 << 20, rounds=4):
    """C3 as BASELINE.json words it: files streamed from the page cache through pinned staging with
    the host-to-device copies of one batch overlapping the scan of the other.  Two batches take
    turns (mi_batch_reset keeps their device memory): while batch A's pipeline runs on the GPU the
    reader threads are already copying batch B's files.  Bounded sample: n_files x file_bytes per
    round in /dev/shm (page-cache warm), `rounds` timed rounds after one warm-up round."""
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="mi_hostfed_", dir=base)
    rng = np.random.default_rng(7)
    paths = []
    try:
        blob = rng.integers(0, 256, file_bytes, dtype=np.uint8)
        for i in range(n_files):
            blob[:8] = np.frombuffer(np.uint64(i).tobytes(), dtype=np.uint8)   # distinct files
            pth = os.path.join(d, "f%04d" % i)
            blob.tofile(pth)
            paths.append(pth)
        half = n_files // 2
        parts = [paths[:half], paths[half:]]
        bs = [eng.batch(len(p), len(p) * file_bytes) for p in parts]

        def fill(k):
            bs[k].reset()
            for i, pth in enumerate(parts[k]):
                bs[k].add_path(pth, file_bytes, i)           # queued: the reader threads do the work

        def one_round():
            # steady state: entering, bs[0] is filled (or filling) and nothing is in flight
            bs[0].submit()                                   # waits for A's bytes, enqueues A's pipeline
            fill(1)                                          # B's reads + H2D run under A's kernels
            bs[1].submit()
            bs[0].wait()
            fill(0)                                          # ... and A's next files under B's kernels
            bs[1].wait()

        fill(0)
        one_round()                                          # warm-up (fresh VRAM, cold threads)
        t0 = time.perf_counter()
        for _ in range(rounds):
            one_round()
        bs[0].submit()                                       # drain the last fill so every byte is counted once
        bs[0].wait()
        dt = time.perf_counter() - t0
        total = (rounds * n_files + len(parts[0])) * file_bytes
        serial = None
        for _ in range(2):                                   # for comparison: one batch, copy THEN scan
            bs[0].reset()
            t1 = time.perf_counter()
            for i, pth in enumerate(parts[0]):
                bs[0].add_path(pth, file_bytes, i)
            bs[0].run()
            serial = len(parts[0]) * file_bytes / (time.perf_counter() - t1)
        for b in bs:
            b.free()
        return {"host_fed_GBps": round(total / dt / 1e9, 2),
                "host_fed_single_batch_GBps": round(serial / 1e9, 2),
                "host_fed_sample": "%d x %d MiB files in %s (page-cache warm) per round, %d rounds + 1 warm-up; "
                                   "mi_batch_add_path -> reader threads + pinned slabs -> H2D; two batches take "
                                   "turns so one batch's copies overlap the other's scan (single_batch = copy, "
                                   "then scan)" % (n_files, file_bytes >> 20, base or "TMPDIR", rounds)}
    finally:
        for pth in paths:
            try:
                os.unlink(pth)
            except OSError:
                pass
        try:
            os.rmdir(d)
        except OSError:
            pass


def make_shard(W, config, args, rank, world, generation):
    """One rank's share of a config; `generation` = which of the in-flight batches (distinct content each)."""
    if config == "c2":
        return W.c2(rank, world, args.files or 100000, generation)
    if config == "c3":
        return W.c3(rank, world, args.files or 1000, generation=generation)
    if config == "c4":
        return W.c4(rank, world, args.files or 1250000, generation)
    if config == "c5u":
        return W.c5u(rank, world, int((args.bytes_per_gpu or 32) * W.GIB), generation)
    return W.c5(rank, world, int((args.bytes_per_gpu or 64) * W.GIB), generation,
                split_threshold=args.split_mib * W.MIB)


def n1_leg(makisu_amd, W, config, args, dev_index, inflight):
    """The N = 1 form of the same per-GPU work, in THIS run on THIS device, before the N-rank job: the config with
    world = 1 (the same files per GPU -- weak scaling), in-batch duplicate marking, no exchange, the same number
    of batches in flight.  `efficiency_vs_n1` = value(N) / (N x this)."""
    import torch
    eng = makisu_amd.Engine(device=dev_index)
    batches, nbytes = [], 0
    for g in range(inflight):
        sh = make_shard(W, config, args, 0, 1, g)
        b = eng.batch(sh.n_files, W.batch_bytes_hint(sh))
        W.fill_batch(b, sh)
        b.run()
        batches.append(b)
        nbytes = sh.n_bytes

    def run(n):
        pending = []
        for k in range(n):
            if len(pending) == inflight:
                batches[pending.pop(0)].wait()
            batches[k % inflight].submit()
            pending.append(k % inflight)
        for i in pending:
            batches[i].wait()

    run(max(1, args.warmup))
    torch.cuda.synchronize(dev_index)
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize(dev_index)
    dt = time.perf_counter() - t0
    for b in batches:
        b.free()
    eng.close()
    return {"value": round(nbytes * args.steps / dt / 2**30, 2), "unit": "GiB/s",
            "ms_per_step": round(dt / args.steps * 1e3, 4), "steps": args.steps, "batches_in_flight": inflight,
            "bytes_per_step": int(nbytes)}


def with_rows_leg(makisu_amd, W, config, args, dev_index, inflight, value_without):
    """The timed steps once more with the RESULTS DELIVERED: a ctx with MI_FLAG_PREFETCH_ROWS (mi_batch_wait brings the packed
    chunk rows to pinned host memory while the other batch in flight keeps the GPU busy), and per step what a host that
    feeds isUpdated and the chunk index reads: mi_batch_chunks_view (64 B per chunk) + mi_batch_files_view (96 B per file),
    both in place.  `value` of the main line leaves the rows on the device; this is the rate with them on the host."""
    import torch
    eng = makisu_amd.Engine(device=dev_index, flags=makisu_amd.FLAG_PREFETCH_ROWS)
    batches, nbytes = [], 0
    for g in range(inflight):
        sh = make_shard(W, config, args, 0, 1, g)
        b = eng.batch(sh.n_files, W.batch_bytes_hint(sh))
        W.fill_batch(b, sh)
        b.run()
        batches.append(b)
        nbytes = sh.n_bytes
    seen = {"rows": 0, "files": 0, "row_bytes": 0, "check": 0}

    def finish(i):
        batches[i].wait()                                       # kernels done, rows already in pinned host memory
        rows = batches[i].chunks_view()
        files = batches[i].files_view()
        seen["rows"], seen["files"] = len(rows), len(files)
        seen["row_bytes"] = rows.nbytes + files.nbytes
        seen["check"] ^= int(rows["length"][-1]) ^ int(files["n_chunks"][0])   # (touched)

    def run(n):
        pending = []
        for k in range(n):
            if len(pending) == inflight:
                finish(pending.pop(0))
            batches[k % inflight].submit()
            pending.append(k % inflight)
        for i in pending:
            finish(i)

    run(max(1, args.warmup))
    torch.cuda.synchronize(dev_index)
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize(dev_index)
    dt = time.perf_counter() - t0
    for b in batches:
        b.free()
    eng.close()
    value = nbytes * args.steps / dt / 2**30
    return {"value": round(value, 2), "unit": "GiB/s", "ms_per_step": round(dt / args.steps * 1e3, 4), "steps": args.steps,
            "batches_in_flight": inflight, "rows_per_step": seen["rows"], "files_per_step": seen["files"],
            "bytes_to_host_per_step": seen["row_bytes"],
            "vs_rows_left_on_device": round(value / value_without, 4) if value_without else None,
            "how": "MI_FLAG_PREFETCH_ROWS: mi_batch_wait packs the chunk rows on the device and copies them to the batch's "
                   "pinned buffers (file rows likewise); per step mi_batch_chunks_view + mi_batch_files_view are read in place by the "
                   "host thread that then submits the next batch"}


def closed_form_check(got, expect):
    """Distinct contents are independent random streams, so chunks of different contents differ -- except the
    shortest ones: a cut candidate on a file's second-to-last byte leaves a 1-BYTE tail chunk (P = 2^-13 per
    file), and there are only 256 of those.  Measured on a C4 shard (1.25 M files, 9 023 970 chunks): 21 such
    coincidences, every one a pair of equal 1-byte chunks.  The closed form therefore holds up to ~1e-5."""
    coincidences = int(expect) - int(got) if got is not None else None
    return {"n_unique": int(got) if got is not None else None, "closed_form": int(expect),
            "short_chunk_coincidences": coincidences,
            "ok": got is not None and 0 <= coincidences <= 2 + int(expect) // 50000}


def reexec_under_torchrun(n, reason):
    """The bare form could not bring its communicator up (or hung doing so): the same job once more as the driver's other
    launch form -- one process per GPU under torch.distributed.run, whose own chain (native communicator per rank, then the
    torch.distributed driver of the same exchange) gets its chance -- so that a first contact with N real GPUs still ends
    in a line, and the line says which path produced it (config.launch_note).  Replaces this process."""
    print("bench.py: %s -- re-executing under torch.distributed.run" % reason, file=sys.stderr, flush=True)
    env = dict(os.environ, MI_BENCH_REEXEC_REASON=reason[:800])
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def watched(what, seconds, fn):
    """fn() on a thread of its own; (result, None), or (None, why) when it raised or is still running after `seconds`
    (0 = no limit).  A call that hangs inside the collective library cannot be cancelled: the caller re-executes."""
    import threading
    box = {}

    def run():
        try:
            box["ok"] = fn()
        except BaseException as e:                              # noqa: BLE001
            box["err"] = "%s failed: %s" % (what, e)
    t = threading.Thread(target=run, daemon=True)
    t0 = time.perf_counter()
    t.start()
    t.join(seconds if seconds and seconds > 0 else None)
    if t.is_alive():
        return None, "%s did not return within %.0f s" % (what, time.perf_counter() - t0)
    if "err" in box:
        return None, box["err"]
    return box.get("ok"), None


def nccl_debug_tail(path, limit=600):
    try:
        txt = open(path, errors="replace").read().strip()
        return txt[-limit:] if txt else None
    except OSError:
        return None


def single_process_job(args):
    """`python bench.py --gpus N` without torchrun: ONE process drives N devices -- the shape a Go host has
    (INTEGRATION.md "Multi-GPU from Go"): one ctx per device, one host thread per device for everything that is
    per-device (generation, submit, wait), mi_comm_init_all once, and per step ONE mi_dedup_allgather_all over
    the N batches (csrc/mi_comm.hip: the N all-gathers in one group, every rank marks its own rows).  No torch in
    the job: torch only answers "how many devices" and synchronises them around the timed region."""
    import concurrent.futures as cf
    import torch
    import makisu_amd
    from makisu_amd import distributed as mdist
    from makisu_amd import workloads as W

    n = args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the hot path)")
    forced = os.environ.get("MI_BENCH_FORCE_DEVICE")          # self-test: all ranks on one GPU (needs the RCCL double)
    if forced is None and torch.cuda.device_count() < n:
        raise SystemExit("--gpus %d but this node shows %d device(s) (WORLD_SIZE is not set: this is the "
                         "single-process form, one ctx per device)" % (n, torch.cuda.device_count()))
    devs = [int(forced)] * n if forced is not None else list(range(n))
    config = args.config if args.config != "auto" else "c4"
    if config == "c3":
        raise SystemExit("c3 is BASELINE.json's one-GPU streaming config; N > 1 runs c4 (default), c5, c5u or c2")
    inflight = args.inflight if args.inflight > 0 else default_inflight(config, True)
    if args.steps <= 0:
        args.steps = {"c2": 20, "c4": 3, "c5": 4, "c5u": 5}[config]
    if args.warmup < 0:
        args.warmup = 3 if config == "c2" else 1
    pool = cf.ThreadPoolExecutor(n)

    def each(fn):                                              # one host thread per device
        return list(pool.map(fn, range(n)))

    # the N = 1 form of the same per-GPU work on EVERY device of the job at once (boxes -- and the GPUs of one box -- differ
    # by a few per cent: efficiency_vs_n1 is against their mean, not against device 0's luck)
    n1 = None
    if not args.no_n1:
        if len(set(devs)) == n:
            legs = each(lambda r: n1_leg(makisu_amd, W, config, args, devs[r], inflight))
            vals = [leg["value"] for leg in legs]
            n1 = dict(legs[0], value=round(float(np.mean(vals)), 2), per_device_value=vals,
                      note="every device ran the config with world = 1 at the same time, one host thread each; value = mean")
        else:                                                  # self-test: the ranks share a device
            n1 = dict(n1_leg(makisu_amd, W, config, args, devs[0], inflight), note="one leg: the ranks of this job share device %d" % devs[0])

    engines = each(lambda r: makisu_amd.Engine(device=devs[r], flags=makisu_amd.FLAG_NO_DEDUP))
    info = engines[0].device_info()
    # first contact: the communicator bring-up and, below, the first exchange run under a watchdog.  What RCCL has to say
    # about a failure (NCCL_DEBUG=WARN, into a file of this job's) travels in the line of whoever finishes the job.
    import tempfile
    dbg_file = os.path.join(tempfile.gettempdir(), "mi_bench_nccl_%d.log" % os.getpid())
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    os.environ.setdefault("NCCL_DEBUG_FILE", dbg_file)

    def give_up(why):
        tail = nccl_debug_tail(os.environ.get("NCCL_DEBUG_FILE", dbg_file))
        reexec_under_torchrun(n, "bare --gpus %d: %s%s" % (n, why, ("; RCCL says: " + tail) if tail else ""))

    _, why = watched("mi_comm_init_all over %d devices" % n, args.watchdog_s, lambda: makisu_amd.comm_init_all(engines))
    if why:
        give_up(why)
    rccl_ranks = [e.comm_ranks() for e in engines]             # ncclCommCount of the library's own communicators
    if rccl_ranks != [n] * n:
        raise SystemExit("the collective library sees %s rank(s), --gpus says %d: no line" % (rccl_ranks, n))
    shards = [[make_shard(W, config, args, r, n, g) for g in range(inflight)] for r in range(n)]

    def build(r):
        out = []
        for sh in shards[r]:
            b = engines[r].batch(sh.n_files, W.batch_bytes_hint(sh))
            out.append((b, W.fill_batch(b, sh)))
        return out
    built = each(build)
    part_rounds = 0
    if any(sh.parts is not None for row in shards for sh in row):
        for g in range(inflight):                              # the parts' owners agree on the boundary cuts, once
            part_rounds = max(part_rounds, mdist.resolve_parts_local([built[r][g] for r in range(n)]))
    batches = [[bk[0] for bk in row] for row in built]
    each(lambda r: [b.run() for b in batches[r]])              # generates the data on the device, first pass

    rec = {"step_ms": [[] for _ in range(n)], "sha_ms": [[] for _ in range(n)], "cdc_ms": [[] for _ in range(n)],
           "gather_ms": [[] for _ in range(n)], "mark_ms": [[] for _ in range(n)], "exchange_host_ms": []}
    checks = {}

    def finish(g, record):
        each(lambda r: batches[r][g].wait())
        t0 = time.perf_counter()
        n_total, n_unique = makisu_amd.dedup_allgather_all([batches[r][g] for r in range(n)], form=args.exchange_form)
        x_ms = (time.perf_counter() - t0) * 1e3
        if record:
            rec["exchange_host_ms"].append(x_ms)
            for r in range(n):
                st = engines[r].stats()
                ga, ma = engines[r].comm_exchange_ms()
                rec["step_ms"][r].append(st["ms_total"])
                rec["sha_ms"][r].append(st["ms_sha_chunks"])
                rec["cdc_ms"][r].append(st["ms_cdc"])
                rec["gather_ms"][r].append(ga)
                rec["mark_ms"][r].append(ma)
            checks[g] = (n_total, n_unique, [engines[r].stats()["n_chunks"] for r in range(n)])

    def run_steps(k_steps, record):
        pending = []
        for k in range(k_steps):
            if len(pending) == inflight:
                finish(pending.pop(0), record)
            g = k % inflight
            each(lambda r: batches[r][g].submit())
            pending.append(g)
        while pending:
            finish(pending.pop(0), record)

    def sync_all():
        for d in sorted(set(devs)):
            torch.cuda.synchronize(d)

    def first_exchange():                                      # one step, alone: submit, wait, the collective, the marking
        each(lambda r: batches[r][0].submit())
        finish(0, False)
    _, why = watched("the first mi_dedup_allgather_all over %d ranks" % n, args.watchdog_s, first_exchange)
    if why:
        give_up(why)
    run_steps(args.warmup, False)
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    sync_all()
    dt = time.perf_counter() - t0

    job_bytes = sum(shards[r][0].n_bytes for r in range(n))
    value = job_bytes * args.steps / dt / 2**30
    # the dominant kernel one batch at a time on rank 0, the other ranks idle: its own duration
    serial_sha, st = [], None
    for i in range(2 + min(7, max(3, args.steps))):
        batches[0][0].submit()
        batches[0][0].wait()
        st = engines[0].stats()
        if i >= 2:
            serial_sha.append(st["ms_sha_chunks"])
    alg = st["bytes_in"] + 52 * st["n_chunks"]
    s_ms = float(np.mean(serial_sha))
    valu_roof = max(engines[0].sha_valu_roof(w, 0) / 1e9 for w in (8, 4))
    # closed form of the job-wide unique count
    g_last = (args.steps - 1) % inflight
    n_total, n_unique, per_rank_chunks = checks[g_last]
    if config in ("c5", "c5u"):
        expect = sum(int(batches[r][g_last].files()["n_chunks"][shards[r][g_last].originals].sum()) for r in range(n))
    else:
        expect = n_total                                      # c2 / c4: every content is distinct
    desc = shards[0][0]
    mean = lambda rows: [round(float(np.mean(x)), 4) for x in rows]     # noqa: E731
    out = {
        "metric": "GiB/s hashed (Gear CDC + SHA-256 per chunk)",
        "value": round(value, 2), "unit": "GiB/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%s (seed 0x%X, device-resident), Gear CDC mask 13 bits / min 2 KiB / max 64 KiB, SHA-256 per "
                               "chunk, per-file chunk root; digest-set all-gather over RCCL inside the library "
                               "(mi_dedup_allgather_all) + job-wide duplicate marking; %d batches in flight per GPU"
                               % (desc.describe, desc.seed, inflight),
                   "name": config, "files_per_gpu": int(desc.n_files), "bytes_per_gpu": int(desc.n_bytes),
                   "job_bytes_per_step": int(job_bytes), "chunks_per_rank_last_batch": [int(x) for x in per_rank_chunks],
                   "parallelism": "files sharded x%d (%s)" % (n, "LPT by bytes" if config in ("c5", "c5u") else "file index mod N"),
                   "launch": "single process, %d ctxs, one host thread per device (mi_comm_init_all)" % n,
                   "batches_in_flight": inflight, "exchange": "native", "exchange_form": args.exchange_form,
                   "rccl_ranks": int(rccl_ranks[0]),
                   "rccl_ranks_per_ctx": [int(x) for x in rccl_ranks],
                   "rccl_library": os.environ.get("MI_RCCL_LIB", "librccl (dlopen)"),
                   "devices": devs, "device": info["name"].strip(), "n_cu": info["n_cu"]},
        "per_rank": {"step_ms": mean(rec["step_ms"]), "cdc_ms": mean(rec["cdc_ms"]), "sha_chunks_ms": mean(rec["sha_ms"]),
                     "exchange_gather_ms": mean(rec["gather_ms"]), "marking_ms": mean(rec["mark_ms"]),
                     "note": "device time per batch from HIP events (step = first kernel start to last kernel end of the "
                             "batch's pipeline, sharing the GPU with the other batch in flight; gather = the slab "
                             "all-gather on the ctx stream; marking = padding squeeze + job-wide marking of the rank's rows)"},
        "exchange_host_ms_avg": round(float(np.mean(rec["exchange_host_ms"])), 4),
        "n1_same_run": n1,
        "efficiency_vs_n1": round(value / (n * n1["value"]), 4) if n1 else None,
        "dedup_check": dict(closed_form_check(n_unique, expect), n_total=int(n_total)),
        "roofline": {"bound": "hbm", "kernel": "sha256_items_kernel (chunk pass)",
                     "achieved": round(alg / (s_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(alg / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": round(s_ms, 4),
                     "valu_roof_GBps": round(valu_roof, 1),
                     "frac_of_valu_roof": round(alg / (s_ms * 1e-3) / 1e9 / valu_roof, 4),
                     "path_frac": round(job_bytes / n * 1.006 / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS, 4),
                     "note": "rank 0's chunk pass one batch at a time right after the timed region (HIP events on the "
                             "batch's stream, the other ranks idle); in the timed region the launches of the batches in "
                             "flight share the SIMDs (per_rank.sha_chunks_ms); path_frac = a rank's whole step, "
                             "algorithmic bytes / ms_per_step / peak"},
    }
    if config in ("c5", "c5u"):
        out["config"].update({"lpt_imbalance_max_over_mean_bytes": round(desc.imbalance, 6),
                              "files_split_into_parts_job": getattr(desc, "n_split_files_job", 0),
                              "part_boundary_rounds": part_rounds})
    for row in batches:
        for b in row:
            b.free()
    for e in engines:
        e.comm_destroy()
        e.close()
    pool.shutdown()
    if not args.no_cpu_baseline:                               # north_star: the host's scanner "in the same run", at N > 1 too
        try:
            out["cpu_baseline"] = cpu_baseline(shards[0][0])
        except Exception as e:                                  # noqa: BLE001
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                           # noqa: BLE001
        pass
    sys.stderr.flush()
    print(json.dumps(out), flush=True)


def main():
    t_main = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="default 20 (c2) / 3-5 for the larger configs")
    ap.add_argument("--warmup", type=int, default=-1)
    ap.add_argument("--config", default="auto", choices=["auto", "c2", "c3", "c4", "c5", "c5u"],
                    help="auto = c2 on one GPU (the config the metric is quoted on), c4 on several")
    ap.add_argument("--files", type=int, default=0, help="files per GPU (c2: 100000, c3: 1000, c4: 1250000)")
    ap.add_argument("--bytes-per-gpu", type=float, default=0, help="c5: GiB per GPU (default 64; c5u 32)")
    ap.add_argument("--split-mib", type=int, default=256,
                    help="c5 with N > 1: files of this many MiB and more are split into one part per GPU")
    ap.add_argument("--inflight", type=int, default=0,
                    help="batches in flight (default 2; 3 for c2 with an exchange; 1 = one batch at a time: every "
                         "kernel then has the GPU to itself in the timed region too -- the form the rocprof kernel "
                         "trace under profiles/ is taken with)")
    ap.add_argument("--no-n1", action="store_true",
                    help="N > 1: skip the N = 1 leg that `efficiency_vs_n1` comes from")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true", help="c3: skip the host-fed leg")
    ap.add_argument("--no-with-rows", action="store_true", help="N = 1: skip the leg that delivers the result rows to the host")
    ap.add_argument("--no-commit-e2e", action="store_true",
                    help="N = 1, c2: skip the commit table (mi_memfs_commit_layer with and without the GPU scan on two trees)")
    ap.add_argument("--watchdog-s", type=float, default=120.0,
                    help="bare --gpus N: seconds the communicator bring-up and the first exchange may take before the job is "
                         "re-executed under torch.distributed.run (0 = wait for ever)")
    ap.add_argument("--backend", default="auto",
                    help="torch.distributed backend under torchrun: auto = gloo with --exchange native (torch then "
                         "only ships the id and runs the host-side barrier and reductions -- the one RCCL "
                         "communicator in the process is the library's), nccl (= RCCL) with --exchange torch")
    ap.add_argument("--exchange", default="native", choices=["torch", "native"],
                    help="who runs the digest all-gather: the library's own RCCL binding (default: "
                         "mi_dedup_allgather / mi_dedup_allgather_all, what a Go host uses) or torch.distributed")
    ap.add_argument("--exchange-form", default="allgather", choices=["allgather", "alltoall"],
                    help="with --exchange native: the digest all-gather + range marking (default), or the hash-partitioned "
                         "form -- every digest to ONE owner rank and 8 bytes back (mi_dedup_alltoall: same results; 38.5 "
                         "bytes per row over xGMI at 8 ranks instead of 224)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the digest exchange + global marking even with one rank (self-test)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world == 1 and args.gpus > 1:
        return single_process_job(args)                      # launched bare: one process, N ctxs

    import torch
    import torch.distributed as dist
    import makisu_amd
    from makisu_amd import distributed as mdist
    from makisu_amd import workloads as W

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the hot path)")
    dev_index = int(os.environ.get("MI_BENCH_FORCE_DEVICE", local_rank))   # self-test: all ranks on one GPU
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    config = args.config if args.config != "auto" else ("c2" if world == 1 else "c4")
    exchange = world > 1 or args.force_exchange
    if args.inflight <= 0:
        args.inflight = default_inflight(config, exchange)
    if args.steps <= 0:
        args.steps = {"c2": 20, "c3": 3, "c4": 3, "c5": 4, "c5u": 5}[config]
    if args.warmup < 0:
        args.warmup = 3 if config == "c2" else 1
    backend = args.backend
    if backend == "auto":
        backend = "gloo" if args.exchange == "native" else "nccl"
    if exchange or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = device if backend == "nccl" else torch.device("cpu")   # small host-side reductions

    def all_ranks(v):                                         # a python value of every rank, rank order
        if not (dist.is_initialized() and world > 1):
            return [v]
        out = [None] * world
        dist.all_gather_object(out, v)
        return out

    # the N = 1 form of the same per-GPU work, every rank on its own GPU at the same time, before the job
    n1 = None
    if world > 1 and not args.no_n1 and config != "c3":
        n1_mine = n1_leg(makisu_amd, W, config, args, dev_index, args.inflight)
        vals = all_ranks(n1_mine["value"])
        n1 = dict(n1_mine, value=round(float(np.mean(vals)), 2), per_rank_value=vals,
                  note="every rank ran the config with world = 1 on its own GPU at the same time; value = mean")

    # with more than one rank the global marking after the all-gather supersedes the in-batch one
    eng = makisu_amd.Engine(device=dev_index,
                            flags=makisu_amd.FLAG_NO_DEDUP if exchange else 0)
    info = eng.device_info()
    exchange_note = None
    if exchange and args.exchange == "native":
        # torchrun only ships the 128-byte id.  Should the library's communicator not come up on some rank
        # (it has met real RCCL with more than one rank on few machines), every rank falls back to the torch
        # driver of the same exchange and the line says so -- a first contact must still yield a line.
        err = None
        dbg = (lambda m: print("bench.py[%d]: %s" % (rank, m), file=sys.stderr, flush=True)) if os.environ.get("MI_BENCH_DEBUG") else (lambda m: None)
        try:
            uid = [None]
            if rank == 0:
                try:
                    uid[0] = eng.comm_unique_id()
                except Exception as e:                          # noqa: BLE001  (the peers wait for SOMETHING from rank 0)
                    uid[0] = ("no id", str(e))
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            dbg("id shipped")
            if isinstance(uid[0], tuple):
                raise RuntimeError("rank 0 could not create the communicator id: %s" % uid[0][1])
            # (under the same watchdog as the bare form's bring-up: a rank that hangs in ncclCommInitRank reports it, and all
            #  ranks fall back together -- the torch driver of the same exchange gets its chance)
            _, why = watched("mi_comm_init_rank(%d of %d)" % (rank, world), args.watchdog_s, lambda: eng.comm_init_rank(world, rank, uid[0]))
            if why:
                raise RuntimeError(why)
            dbg("communicator up")
            if os.environ.get("MI_BENCH_FAIL_NATIVE_ON_RANK") == str(rank):      # self-test of the fallback
                raise RuntimeError("simulated failure after the communicator came up")
        except Exception as e:                                  # noqa: BLE001
            err = "rank %d: %s" % (rank, e)
        dbg("err = %r" % (err,))
        errs = [e for e in all_ranks(err) if e]
        dbg("all ranks heard: %r" % (errs,))
        if errs:
            try:
                if eng.comm_ranks():
                    eng.comm_destroy()
            except Exception:                                   # noqa: BLE001
                pass
            dbg("communicator dropped")
            args.exchange = "torch"
            exchange_note = "native communicator failed (%s): torch.distributed drives the exchange" % "; ".join(errs)[:400]
            print("bench.py: " + exchange_note, file=sys.stderr)
    torch_group = None
    if exchange and args.exchange == "torch" and backend != "nccl" and args.backend == "auto" and world > 1:
        torch_group = dist.new_group(backend="nccl")            # the fallback's RCCL group
    # how many ranks the collective library itself sees -- a scaling line must be able to prove its N
    rccl_ranks = None
    if exchange:
        if args.exchange == "native":
            rccl_ranks = eng.comm_ranks()                    # ncclCommCount of the library's own communicator
        elif backend == "nccl" or torch_group is not None:
            rccl_ranks = dist.get_world_size(torch_group)    # torch's NCCL(=RCCL) group
        if rccl_ranks is not None and rccl_ranks != args.gpus and (world > 1 or args.force_exchange):
            raise SystemExit("the collective library sees %d rank(s), --gpus says %d: no line" % (rccl_ranks, args.gpus))

    # The rank's share of the config, as `inflight` batches of distinct content.  Step k runs on
    # batch k % inflight: every step is a complete pass (all outputs recomputed); a step is
    # submitted while the previous one is still running so the Gear pass of one overlaps the SHA
    # pass of the other.  c3 cuts the rank's 1000 files into `inflight` parts that
    # together make ONE step (125 GiB fits HBM once, not twice).
    def make(generation):
        return make_shard(W, config, args, rank, world, generation)

    # c3's host-fed leg runs FIRST: right after the resident batches are freed the driver is still
    # wiping their 130 GB of VRAM on the SDMA engines the H2D copies need (measured: 18 GB/s then,
    # 46 GB/s on a quiet device)
    host_fed = None
    if config == "c3" and rank == 0 and world == 1 and not args.no_host_fed:
        host_fed = host_fed_rate(eng)

    split = config == "c3"
    shards = []
    if split:
        whole = make(0)
        for part in np.array_split(np.arange(whole.n_files), args.inflight):
            shards.append(W.Shard(whole.name, whole.seed, whole.sizes[part], whole.cids[part],
                                  whole.global_index[part], whole.n_global_files, describe=whole.describe))
        step_bytes = whole.n_bytes
        desc_shard = whole
    else:
        shards = [make(i) for i in range(args.inflight)]
        step_bytes = shards[0].n_bytes
        desc_shard = shards[0]
    batches = []
    part_rounds = 0
    for sh in shards:
        b = eng.batch(sh.n_files, W.batch_bytes_hint(sh))
        keys = W.fill_batch(b, sh)
        if config in ("c5", "c5u") and world > 1:
            # the parts' owners agree on the cuts at the part boundaries (8 bytes per boundary over the
            # host group; every rank calls, also one that owns no part); later steps reuse the entries
            part_rounds = max(part_rounds, mdist.resolve_parts(b, keys))
        b.run()                                   # generates the data on the device, first pass
        batches.append(b)
    launches_per_step = len(batches) if split else 1
    sha_ms, sha_alg_bytes, stats_sum, checks = [], [], {}, {}
    xrec = {"host_ms": [], "gather_ms": [], "mark_ms": [], "step_ms": []}

    def finish(i, record):
        b = batches[i]
        b.wait()
        n_unique = None
        t_x = time.perf_counter()
        if exchange:
            if args.exchange == "native":
                _, n_unique, _ = (b.dedup_alltoall if args.exchange_form == "alltoall" else b.dedup_allgather)()   # RCCL inside the library
            else:
                _, n_unique, _, _ = mdist.global_dedup(eng, b, device, group=torch_group)
        if record:
            st = eng.stats()
            sha_ms.append(st["ms_sha_chunks"])
            sha_alg_bytes.append(st["bytes_in"] + 52 * st["n_chunks"])
            for k, v in st.items():
                if k.startswith("ms_"):
                    stats_sum[k] = stats_sum.get(k, 0.0) + v
            checks[i] = (st["n_chunks"], n_unique if exchange else st["n_unique"])
            xrec["step_ms"].append(st["ms_total"])
            if exchange:
                xrec["host_ms"].append((time.perf_counter() - t_x) * 1e3)
                if args.exchange == "native":
                    ga, ma = eng.comm_exchange_ms()
                    xrec["gather_ms"].append(ga)
                    xrec["mark_ms"].append(ma)

    def run_steps(n, record, inflight=None):
        inflight = inflight or args.inflight
        pending = []
        for k in range(n * launches_per_step):
            if len(pending) == inflight:
                finish(pending.pop(0), record)
            i = k % len(batches)
            batches[i].submit()
            pending.append(i)
        while pending:
            finish(pending.pop(0), record)

    def fence():
        torch.cuda.synchronize(device)
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize(device)

    try:
        bus = torch.cuda.get_device_properties(dev_index).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(dev_index), "pci_domain_id", 0)
        pci = "%04x:%02x:%02x.0" % (dom, bus, torch.cuda.get_device_properties(dev_index).pci_device_id)
    except Exception:                                           # noqa: BLE001
        pci = None
    sampler = ClockSampler(pci)
    if sampler.freq is None and pci is not None:
        sampler = ClockSampler(None)                            # containers often show one card only
    run_steps(args.warmup, False)
    fence()
    sampler.start()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    fence()
    dt = time.perf_counter() - t0
    clocks = sampler.stop() if rank == 0 else None
    chunks_last_batch = int(eng.stats()["n_chunks"])            # of the timed region's last step
    if dist.is_initialized() and world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    job_bytes = step_bytes
    if dist.is_initialized() and world > 1:             # shards differ in size for c5
        t = torch.tensor([step_bytes], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        job_bytes = int(t.item())
    else:
        job_bytes = step_bytes * world

    # Right behind the timed region: the SAME steps ONE batch at a time (no exchange), so that every kernel has
    # the GPU to itself and the event-bracketed duration of a launch is the kernel's own -- `roofline` is taken
    # from these launches when the timed region keeps several batches in flight.  2 untimed steps (the clock
    # ramps up again after the host-side pause), then --steps timed ones.
    serial_sha_ms, serial_alg, serial_phase, one_at_a_time = [], [], {}, None
    if args.inflight > 1:
        torch.cuda.synchronize(device)
        n_serial = args.steps * launches_per_step
        acc = {}
        for i in range(2 * launches_per_step + n_serial):
            if i == 2 * launches_per_step:
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
            b = batches[i % len(batches)]
            b.submit()
            b.wait()
            if i >= 2 * launches_per_step:
                st = eng.stats()
                serial_sha_ms.append(st["ms_sha_chunks"])
                serial_alg.append(st["bytes_in"] + 52 * st["n_chunks"])
                for k, v in st.items():
                    if k.startswith("ms_"):
                        acc[k] = acc.get(k, 0.0) + v
        torch.cuda.synchronize(device)
        dt1 = time.perf_counter() - t1
        serial_phase = {k: round(v / n_serial, 4) for k, v in sorted(acc.items())}
        if not exchange:
            one_at_a_time = {"value": round(step_bytes * args.steps / dt1 / 2**30, 2), "unit": "GiB/s",
                             "ms_per_step": round(dt1 / args.steps * 1e3, 4), "steps": args.steps,
                             "note": "the same steps one batch at a time, right after the timed region: round 3's "
                                     "form of the headline (BENCH_r03: 1053.77 GiB/s, 5.7921 ms)"}
    # the SHA-256 VALU roof of THIS device in THIS run, right behind the serial steps (same thermal and
    # power state): best of three 10 ms launches of the compression alone
    sampler.start()
    roofs = {w: eng.sha_valu_roof(w, 0) / 1e9 for w in (8, 4)}   # waves per SIMD: the denser form draws more power
    valu_roof = max(roofs.values())
    roof_clocks = sampler.stop() if rank == 0 else None

    # The same pass with the OTHER load scheme (quad-cooperative: far fewer address translations), a few
    # serial steps on a second ctx: on a box whose lane-owned launches run well below the VALU roof this
    # tells a translation-bound kernel (cooperative faster) from a clock-bound one (both slow alike).
    other_scheme = None
    if config == "c2" and world == 1 and rank == 0:
        auto_coop = shards[0].n_bytes >= (9 << 30)
        e2 = makisu_amd.Engine(device=dev_index, sha_load_scheme=makisu_amd.SHA_LOADS_LANE if auto_coop
                               else makisu_amd.SHA_LOADS_COOP)
        b2 = e2.batch(shards[0].n_files, shards[0].n_bytes)
        b2.add_synthetic(shards[0].sizes, shards[0].cids, seed=shards[0].seed)
        ms2 = []
        for i in range(7):
            b2.run() if i == 0 else b2.rerun()
            if i >= 2:
                ms2.append(e2.stats()["ms_sha_chunks"])
        b2.free()
        e2.close()
        other_scheme = {"scheme": "lane-owned" if auto_coop else "quad-cooperative",
                        "serial_launch_ms": round(float(np.median(ms2)), 4)}

    # closed-form check of the duplicate marking (c5: 90 % duplicate files): the job-wide unique
    # count must equal the chunk count of the first file of every distinct content
    dedup_check = None
    if not split and (exchange or world == 1):
        if config in ("c5", "c5u"):
            files0 = batches[0].files()
            expect = int(files0["n_chunks"][shards[0].originals].sum())
        else:
            expect = checks[0][0]                      # c2 / c4: every content is distinct
        if dist.is_initialized() and world > 1:
            t = torch.tensor([expect], dtype=torch.int64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            expect = int(t.item())
        dedup_check = closed_form_check(checks[0][1], expect)

    value = job_bytes * args.steps / dt / 2**30
    # dominant kernel: SHA-256 per chunk.  Algorithmic bytes per launch: every file byte read
    # once + 32 B digest written per chunk + the 20 B queue descriptor read per chunk.
    span_ms = float(np.mean(sha_ms))                             # timed region: shared with the other batches in flight
    if serial_sha_ms:
        alg_bytes, sha_avg_ms = float(np.mean(serial_alg)), float(np.mean(serial_sha_ms))
    else:
        alg_bytes, sha_avg_ms = float(np.mean(sha_alg_bytes)), span_ms
    achieved = alg_bytes / (sha_avg_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if config == "c2" and os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            traffic = tj.get("sha256_items_kernel_bytes_per_launch")
            traffic_src = tj.get("source")
        except Exception:
            traffic = None
    mode = ("%d batches in flight (value, ms_per_step); roofline from the one-batch-at-a-time steps of the same run"
            % args.inflight) if args.inflight > 1 else "one batch at a time (value, ms_per_step and roofline alike)"
    out = {
        "metric": "GiB/s hashed (Gear CDC + SHA-256 per chunk)",
        "value": round(value, 2), "unit": "GiB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%s (seed 0x%X, device-resident), Gear CDC mask 13 bits / min 2 KiB / "
                               "max 64 KiB, SHA-256 per chunk, per-file chunk root, duplicate marking%s; %s"
                               % (desc_shard.describe, desc_shard.seed,
                                  "; digest-set all-gather over RCCL (%s) + job-wide marking" % args.exchange
                                  if exchange else "", mode),
                   "name": config, "files_per_gpu": int(sum(s.n_files for s in shards) if split else shards[0].n_files),
                   "bytes_per_gpu": int(step_bytes), "job_bytes_per_step": int(job_bytes),
                   "chunks_last_batch": chunks_last_batch,
                   "parallelism": "files sharded x%d (%s)" % (world, "LPT by bytes" if config in ("c5", "c5u") else "file index mod N"),
                   "launch": "one process per GPU" if world > 1 else "one process, one GPU",
                   "batches_in_flight": args.inflight, "launches_per_step": launches_per_step,
                   "exchange": (args.exchange if exchange else None),
                   "exchange_form": (args.exchange_form if exchange and args.exchange == "native" else None),
                   "rccl_ranks": rccl_ranks,
                   "exchange_backend": (("library RCCL binding; torch.distributed %s for the id, barrier and scalars" % backend)
                                        if exchange and args.exchange == "native" else
                                        ("nccl" if torch_group is not None else backend) if exchange else None),
                   "device": info["name"].strip(), "n_cu": info["n_cu"],
                   "results": "left on the device (16 B of counts read back per batch); mi_batch_chunks_view / "
                              "mi_batch_files bring 64 B per chunk + 96 B per file to the host on demand "
                              "(one packed copy, ~1.6 ms per C2 batch), outside this metric"},
        "roofline": {"bound": "hbm", "kernel": "sha256_items_kernel (chunk pass)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "traffic_from_profile_run": traffic_src,
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     "avg_launch_ms": round(sha_avg_ms, 4),
                     "avg_launch_ms_from": ("the %d one-batch-at-a-time steps right after the timed region" % len(serial_sha_ms))
                                           if serial_sha_ms else "the timed region",
                     "valu_roof_GBps": round(valu_roof, 1),
                     "valu_roof_source": "mi_sha_valu_roof in this run, right after the timed region: the 64-round "
                                         "compression alone on every SIMD (no memory traffic), best of 3 launches at 8 and "
                                         "at 4 waves per SIMD: %s GB/s.  The roof launch is pure VALU work and is itself "
                                         "subject to the socket's power management: on a box where it is throttled the "
                                         "hashing pass can sit at (or a few %% above) it -- both are then at the same limit"
                                         % {k: round(v, 1) for k, v in roofs.items()},
                     "frac_of_valu_roof": round(achieved / valu_roof, 4),
                     "path_frac": round(job_bytes / world * 1.006 / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS, 4),
                     "note": "SHA-256 is integer-VALU bound on CDNA4 (valu_roof_GBps, measured in this run); "
                             "the HBM fraction cannot exceed valu_roof / peak. "
                             "achieved/avg_launch_ms: HIP events around the launch on the batch's stream, %s; "
                             "path_frac = whole CDC+SHA step per GPU, algorithmic bytes / ms_per_step / peak"
                             % ("one batch at a time in the timed region, the kernel has the GPU to itself" if args.inflight == 1 else
                                "from the one-batch-at-a-time steps of this run; in the timed region the kernel shares the "
                                "SIMDs with the other in-flight batches' passes and the span between its events is "
                                "inflight_span_ms")},
        "phase_ms_avg": {k: round(v / max(1, len(sha_ms)), 4) for k, v in sorted(stats_sum.items())},
        "phase_note": "per-batch stream timelines of the timed region; with several batches in flight a phase's span "
                      "includes time it shared the GPU with the other batches; serial_phase_ms = one batch at a time",
    }
    if args.inflight == 1:
        out["serial_phase_ms"] = dict(out["phase_ms_avg"])      # one batch at a time: the timed region IS serial
    else:
        out["serial_phase_ms"] = serial_phase
        out["roofline"]["inflight_span_ms"] = round(span_ms, 4)
    if one_at_a_time:
        out["one_batch_at_a_time"] = one_at_a_time
    if exchange_note:
        out["config"]["exchange_note"] = exchange_note
    if os.environ.get("MI_BENCH_REEXEC_REASON"):
        out["config"]["launch_note"] = ("re-executed by the bare single-process form after: " + os.environ["MI_BENCH_REEXEC_REASON"])
    if config in ("c5", "c5u"):
        out["config"].update({"lpt_imbalance_max_over_mean_bytes": round(desc_shard.imbalance, 6),
                              "files_split_into_parts_job": getattr(desc_shard, "n_split_files_job", 0),
                              "parts_this_rank": sum(1 for p_ in (desc_shard.parts or ()) if p_[3] >= 0),
                              "part_boundary_rounds": part_rounds})
    if exchange:
        mine = {"step_ms": round(float(np.mean(xrec["step_ms"])), 4),
                "exchange_host_ms": round(float(np.mean(xrec["host_ms"])), 4),
                "exchange_gather_ms": round(float(np.mean(xrec["gather_ms"])), 4) if xrec["gather_ms"] else None,
                "marking_ms": round(float(np.mean(xrec["mark_ms"])), 4) if xrec["mark_ms"] else None}
        rows = all_ranks(mine)
        out["per_rank"] = {k: [r_[k] for r_ in rows] for k in mine}
        out["per_rank"]["note"] = ("step = device time of a batch's pipeline (first kernel start to last kernel end, sharing "
                                   "the GPU with the other batch in flight); exchange_host = wall time of the exchange call; "
                                   "gather / marking = device time of the slab all-gather and of the job-wide marking "
                                   "(HIP events on the ctx stream, native exchange only)")
    if world > 1:
        out["n1_same_run"] = n1
        out["efficiency_vs_n1"] = round(value / (world * n1["value"]), 4) if n1 else None
    if clocks is not None:
        out["clocks"] = {"timed_region": clocks, "valu_roof_launches": roof_clocks}
    if dedup_check:
        out["dedup_check"] = dedup_check
    if other_scheme:
        out["roofline"]["other_load_scheme_serial"] = other_scheme
    for b in batches:
        b.free()
    batches = []
    if rank == 0 and world == 1:
        if host_fed:
            out["config"].update(host_fed)
        # the legs behind the headline: one that fails (a full /dev/shm, a host without room for the sample) says so in its
        # place -- the measured line still comes out
        def leg(name, fn):
            try:
                out[name] = fn()
            except Exception as e:                                  # noqa: BLE001
                out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                print("bench.py: the %s leg failed: %s" % (name, e), file=sys.stderr)
        if not args.no_with_rows and not exchange and config in ("c2", "c4", "c5", "c5u"):
            leg("with_rows_on_host", lambda: with_rows_leg(makisu_amd, W, config, args, dev_index, args.inflight, value))
        if not args.no_commit_e2e and config == "c2" and not exchange:
            # what the GPU buys (and costs) a BUILD: step.commitLayer end to end on two trees, with the scan inside and
            # without (tools/commit_layer_bench.py says what each row is)
            def commit_table():
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from commit_layer_bench import commit_e2e
                # a ctx's FIRST host-fed commit starts its reader threads (a pinned slab each), the read-back windows and the arena's
                # address range: once per process, 0.05-0.15 s (profiles/r06_first_commit_order.txt).  It is paid here, by a commit
                # of 64 small files whose time is reported, so that the tables below compare commits, not who came first.
                t0 = time.perf_counter()
                eng.warm()                                   # mi_ctx_warm: what a host does beside its own start-up work
                warm = time.perf_counter() - t0
                t0 = time.perf_counter()
                commit_e2e(eng, 64, 65536)
                first = time.perf_counter() - t0
                return {"call": "mi_memfs_commit_layer(fs, ctx | NULL, must_scan = 1, ...), gzip leg off; s_total = wall seconds around the python harness's call "
                                "(the layer's entries stay in the library: no per-entry python work in the timed call), s_call = the library's own clock around the C call",
                        "ctx_warm_s": round(warm, 4),
                        "ctx_warm": "mi_ctx_warm on the ctx the headline ran on (its kernels are loaded: this is the reader threads and their pinned slabs)",
                        "first_use_s": round(first, 4),
                        "first_use": "nine commits of a 64 x 64 KiB tree before the tables (the same three sides, three commits each): the ctx's first "
                                     "host-fed use -- reader threads, pinned slabs, read-back windows -- is in this number, not in the tables",
                        "small_files": commit_e2e(eng, 100000, 4096, all_new_rounds=3), "large_files": commit_e2e(eng, 48, 128 << 20)}
            leg("commit_e2e", commit_table)
        if not args.no_cpu_baseline:
            leg("cpu_baseline", lambda: cpu_baseline(desc_shard))
        # The driver's record keeps `config`, `roofline` and `cpu_baseline` in full and every other key as a name:
        # the product's numbers -- a commit with the GPU scan inside beside the header-only commit it replaces -- go where they
        # survive, in a few hundred bytes; the full tables stay under `commit_e2e` / `with_rows_on_host`.
        try:
            ce = out.get("commit_e2e") or {}
            if "small_files" in ce and "large_files" in ce and isinstance(out.get("cpu_baseline"), dict):
                def row(tree, i):
                    r = ce[tree]["commits"][i]
                    return [r["gpu"]["s_total"], r["gpu_trust_ctime"]["s_total"], r["cpu_header_only"]["s_total"]]
                big = ce["large_files"]["commits"][0]
                # the small tree's "all new" (0.3 s, one sample moves by 15 % with the host): each side's best of its rounds on fresh handles
                rs = ce["small_files"]["commits"][0].get("all_new_rounds_s")
                small_new = [min(rs[k]) for k in ("gpu", "gpu_trust_ctime", "cpu_header_only")] if rs else row("small_files", 0)
                out["cpu_baseline"]["commit_s"] = {
                    "order": "[gpu ctx, gpu ctx + MI_MEMFS_TRUST_CTIME, header-only (ctx NULL = the reference's commit)] wall s; small = 100000 x 4 KiB%s, "
                             "large = 48 x 128 MiB, page cache, gzip off" % (" (all new: each side's best of %d fresh handles)" % len(rs["gpu"]) if rs else ""),
                    "all_new": {"small": small_new, "large": row("large_files", 0)},
                    "nothing_changed": {"small": row("small_files", 1), "large": row("large_files", 1)},
                    "changed_0p1pct": {"small": row("small_files", 2), "large": row("large_files", 2)},
                    "all_new_gpu_over_header_only": {"small": round(small_new[0] / small_new[2], 4),
                                                     "large": round(row("large_files", 0)[0] / row("large_files", 0)[2], 4)},
                    "large_all_new_verified": [big["gpu"].get("files_verified"), big["gpu"].get("chunks_refetched"), big["gpu"].get("arena_moves")]}
            wr = out.get("with_rows_on_host") or {}
            if "vs_rows_left_on_device" in wr:
                out["config"]["with_rows_ratio"] = wr["vs_rows_left_on_device"]
        except Exception as e:                                      # noqa: BLE001
            print("bench.py: the summary for the driver's record failed: %s" % e, file=sys.stderr)
    if exchange and args.exchange == "native":
        eng.comm_destroy()
    eng.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio; flush it first so the JSON line
        # is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stderr.flush()
        # everything this process did since main() began -- imports of torch and the library apart: the legs behind the headline
        # (cpu_baseline, with_rows_on_host, commit_e2e) are most of it; the timed region is steps x ms_per_step
        out["config"]["bench_wall_s"] = round(time.perf_counter() - t_main, 1)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
