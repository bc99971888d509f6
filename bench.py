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
    """Batches in flight when --inflight is not given.
    Without an exchange: ONE batch at a time.  Two in flight are 3-6 % faster (the second batch's passes fill the first
    one's tails -- reported as `two_batches_in_flight`), but their persistent hashing grids then share the SIMDs for most
    of their lives and the span of a launch (6-8 ms between its events, in the bench and in rocprof alike) says nothing
    about the kernel (4.2 ms).  With an exchange there is host-synchronised work to hide: two batches, three for the
    small config."""
    if not exchange:
        return 1
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


def main():
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
                    help="batches in flight (default: 1 = one batch at a time on one GPU without an exchange, "
                         "where every kernel then has the GPU to itself and its event-bracketed duration is its "
                         "own; 2 with an exchange to hide, 3 for c2 with an exchange)")
    ap.add_argument("--no-inflight-extra", action="store_true",
                    help="c2, one GPU: skip the extra region with two batches in flight")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true", help="c3: skip the host-fed leg")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only for the single-GPU "
                         "self-test of the N>1 logic)")
    ap.add_argument("--exchange", default="torch", choices=["torch", "native"],
                    help="who runs the digest all-gather: torch.distributed (default) or the library's "
                         "own RCCL binding, mi_dedup_allgather (what a Go host uses)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the digest exchange + global marking even with one rank (self-test)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import makisu_amd
    from makisu_amd import distributed as mdist
    from makisu_amd import workloads as W

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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
    if exchange or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # with more than one rank the global marking after the all-gather supersedes the in-batch one
    eng = makisu_amd.Engine(device=dev_index,
                            flags=makisu_amd.FLAG_NO_DEDUP if exchange else 0)
    info = eng.device_info()
    if exchange and args.exchange == "native":
        uid = [eng.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)          # torchrun only ships the 128-byte id
        eng.comm_init_rank(world, rank, uid[0])
    # how many ranks the collective library itself sees -- a scaling line must be able to prove its N
    rccl_ranks = None
    if exchange:
        if args.exchange == "native":
            rccl_ranks = eng.comm_ranks()                    # ncclCommCount of the library's own communicator
        elif args.backend == "nccl":
            rccl_ranks = dist.get_world_size()               # torch's NCCL(=RCCL) group, created with device_id above
        if rccl_ranks is not None and rccl_ranks != args.gpus and (world > 1 or args.force_exchange):
            raise SystemExit("the collective library sees %d rank(s), --gpus says %d" % (rccl_ranks, args.gpus))

    # The rank's share of the config, as `inflight` batches of distinct content.  Step k runs on
    # batch k % inflight: every step is a complete pass (all outputs recomputed); a step is
    # submitted while the previous one is still running so the Gear pass of one overlaps the SHA
    # pass of the other.  c3 cuts the rank's 1000 files into `inflight` parts that
    # together make ONE step (125 GiB fits HBM once, not twice).
    def make(generation):
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
        b = eng.batch(sh.n_files, sh.n_bytes + (sh.n_files + 8) * 4096 + (len(sh.parts or ()) << 19))
        if sh.parts is None:
            b.add_synthetic(sh.sizes, sh.cids, seed=sh.seed)
        else:
            # c5 on several GPUs: runs of whole files, and parts of the files that were split
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
        if config in ("c5", "c5u") and world > 1:
            # the parts' owners agree on the cuts at the part boundaries (8 bytes per boundary over the
            # host group; every rank calls, also one that owns no part); later steps reuse the entries
            part_rounds = max(part_rounds, mdist.resolve_parts(b, keys if sh.parts is not None else []))
        b.run()                                   # generates the data on the device, first pass
        batches.append(b)
    launches_per_step = len(batches) if split else 1
    sha_ms, sha_alg_bytes, stats_sum, checks = [], [], {}, {}

    def finish(i, record):
        b = batches[i]
        b.wait()
        n_unique = None
        if exchange:
            if args.exchange == "native":
                _, n_unique, _ = b.dedup_allgather()      # RCCL inside the library
            else:
                _, n_unique, _, _ = mdist.global_dedup(eng, b, device)
        if record:
            st = eng.stats()
            sha_ms.append(st["ms_sha_chunks"])
            sha_alg_bytes.append(st["bytes_in"] + 52 * st["n_chunks"])
            for k, v in st.items():
                if k.startswith("ms_"):
                    stats_sum[k] = stats_sum.get(k, 0.0) + v
            checks[i] = (st["n_chunks"], n_unique if exchange else st["n_unique"])

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

    red_dev = device if args.backend == "nccl" else torch.device("cpu")   # small host-side reductions

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
    # the SHA-256 VALU roof of THIS device in THIS run, right behind the timed region (same thermal and
    # power state): best of three 10 ms launches of the compression alone
    sampler.start()
    roofs = {w: eng.sha_valu_roof(w, 0) / 1e9 for w in (8, 4)}   # waves per SIMD: the denser form draws more power
    valu_roof = max(roofs.values())
    roof_clocks = sampler.stop() if rank == 0 else None
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

    # Outside the timed region: a few SERIAL steps (one batch at a time) so the dominant
    # kernel's launch duration can also be read without another batch sharing the GPU.
    serial_sha_ms, serial_phase = [], {}
    if args.inflight > 1:
        torch.cuda.synchronize(device)
        for i in range(9):                                   # 2 untimed (the clock ramps up again after the
            batches[0].submit()                              # host-side pause) + the median of 7
            batches[0].wait()
            st = eng.stats()
            if i >= 2:
                serial_sha_ms.append(st["ms_sha_chunks"])
        serial_phase = {k: round(v, 4) for k, v in st.items() if k.startswith("ms_")}
        serial_alg = st["bytes_in"] + 52 * st["n_chunks"]

    # c2 on one GPU: the same steps with TWO batches in flight (a second batch of distinct content), outside
    # the timed region -- what a host that always has the next layer ready gets.
    two_in_flight = None
    if config == "c2" and world == 1 and not exchange and args.inflight == 1 and not args.no_inflight_extra:
        sh2 = make(1)
        b_extra = eng.batch(sh2.n_files, sh2.n_bytes + (sh2.n_files + 8) * 4096)
        b_extra.add_synthetic(sh2.sizes, sh2.cids, seed=sh2.seed)
        b_extra.run()
        batches.append(b_extra)
        keep = (list(sha_ms), list(sha_alg_bytes), dict(stats_sum), dict(checks))
        del sha_ms[:], sha_alg_bytes[:]
        run_steps(args.warmup, False, inflight=2)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        run_steps(args.steps, True, inflight=2)
        torch.cuda.synchronize(device)
        dt2 = time.perf_counter() - t1
        two_in_flight = {"value": round(step_bytes * args.steps / dt2 / 2**30, 2), "unit": "GiB/s",
                         "ms_per_step": round(dt2 / args.steps * 1e3, 4), "steps": args.steps,
                         "sha_chunk_pass_span_ms": round(float(np.mean(sha_ms)), 4),
                         "note": "same steps, a second batch of distinct content submitted while the first runs; "
                                 "the span of a hashing launch now includes the time it shares the SIMDs with the "
                                 "other batch's passes"}
        sha_ms[:], sha_alg_bytes[:] = keep[0], keep[1]
        stats_sum.clear(); stats_sum.update(keep[2])
        checks.clear(); checks.update(keep[3])
        batches.pop()
        b_extra.free()

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
        got = checks[0][1]
        # Distinct contents are independent random streams, so chunks of different contents differ --
        # except the shortest ones: a cut candidate on a file's second-to-last byte leaves a 1-BYTE
        # tail chunk (P = 2^-13 per file), and there are only 256 of those.  Measured on a C4 shard
        # (1.25 M files, 9 023 970 chunks): 21 such coincidences, every one a pair of equal 1-byte
        # chunks (tools/debug_c4_dups.py).  The closed form therefore holds up to ~1e-5.
        coincidences = int(expect) - int(got) if got is not None else None
        dedup_check = {"n_unique": int(got) if got is not None else None, "closed_form": int(expect),
                       "short_chunk_coincidences": coincidences,
                       "ok": got is not None and 0 <= coincidences <= 2 + int(expect) // 50000}

    value = job_bytes * args.steps / dt / 2**30
    # dominant kernel: SHA-256 per chunk.  Algorithmic bytes per launch: every file byte read
    # once + 32 B digest written per chunk + the 20 B queue descriptor read per chunk.
    alg_bytes = float(np.mean(sha_alg_bytes))
    sha_avg_ms = float(np.mean(sha_ms))
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
    st = eng.stats()
    out = {
        "metric": "GiB/s hashed (Gear CDC + SHA-256 per chunk)",
        "value": round(value, 2), "unit": "GiB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%s (seed 0x%X, device-resident), Gear CDC mask 13 bits / min 2 KiB / "
                               "max 64 KiB, SHA-256 per chunk, per-file chunk root, duplicate marking%s"
                               % (desc_shard.describe, desc_shard.seed,
                                  "; digest-set all-gather over RCCL (%s) + job-wide marking" % args.exchange
                                  if exchange else ""),
                   "name": config, "files_per_gpu": int(sum(s.n_files for s in shards) if split else shards[0].n_files),
                   "bytes_per_gpu": int(step_bytes), "job_bytes_per_step": int(job_bytes),
                   "chunks_last_batch": chunks_last_batch,
                   "parallelism": "files sharded x%d (%s)" % (world, "LPT by bytes" if config in ("c5", "c5u") else "file index mod N"),
                   "batches_in_flight": args.inflight, "launches_per_step": launches_per_step,
                   "exchange": (args.exchange if exchange else None),
                   "rccl_ranks": rccl_ranks,
                   "exchange_backend": (args.backend if exchange else None),
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
                             "achieved/avg_launch_ms are from the timed region (HIP events around the launch on "
                             "the batch's stream)%s; path_frac = whole CDC+SHA step per GPU, algorithmic "
                             "bytes / ms_per_step / peak"
                             % (": one batch at a time, the kernel has the GPU to itself" if args.inflight == 1 else
                                ", where the kernel shares the GPU with the other in-flight batches' passes; "
                                "serial_* = the same kernel with one batch at a time (median of 7 extra untimed steps)")},
        "phase_ms_avg": {k: round(v / max(1, len(sha_ms)), 4) for k, v in sorted(stats_sum.items())},
        "phase_note": "per-batch stream timelines; with several batches in flight a phase's span "
                      "includes time it shared the GPU with the other batches",
    }
    if args.inflight == 1:
        out["serial_phase_ms"] = dict(out["phase_ms_avg"])      # one batch at a time: the timed region IS serial
    if two_in_flight:
        out["two_batches_in_flight"] = two_in_flight
    if config in ("c5", "c5u"):
        out["config"].update({"lpt_imbalance_max_over_mean_bytes": round(desc_shard.imbalance, 6),
                              "files_split_into_parts_job": getattr(desc_shard, "n_split_files_job", 0),
                              "parts_this_rank": sum(1 for p_ in (desc_shard.parts or ()) if p_[3] >= 0),
                              "part_boundary_rounds": part_rounds})
    if clocks is not None:
        out["clocks"] = {"timed_region": clocks, "valu_roof_launches": roof_clocks}
    if dedup_check:
        out["dedup_check"] = dedup_check
    if serial_sha_ms:
        s_ms = float(np.median(serial_sha_ms))
        s_ach = serial_alg / (s_ms * 1e-3) / 1e9
        out["roofline"].update({"serial_avg_launch_ms": round(s_ms, 4),
                                "serial_achieved": round(s_ach, 1),
                                "serial_frac": round(s_ach / HBM_PEAK_GBPS, 4),
                                "serial_frac_of_valu_roof": round(s_ach / valu_roof, 4)})
        out["serial_phase_ms"] = serial_phase
    if other_scheme:
        out["roofline"]["other_load_scheme_serial"] = other_scheme
    if rank == 0 and world == 1:
        if host_fed:
            out["config"].update(host_fed)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc_shard)
    for b in batches:
        b.free()
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
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
