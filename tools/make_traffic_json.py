#!/usr/bin/env python3
"""profiles/traffic_latest.json from the round's two PMC summaries (FETCH_SIZE, WRITE_SIZE passes of
tools/round_profiles.sh) and its bench line: per-launch HBM bytes of the two hot kernels with the
gfx950 corrections MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 for wide coalesced reads; WRITE_SIZE
calibrated 1:1 on synth_fill_kernel, whose written byte count is known).

  python tools/make_traffic_json.py r02 gpurun_out/round > profiles/traffic_latest.json
"""
import json
import re
import sys


def counters(path, name):
    out = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+%s\s+(\d+)\s+([0-9.]+)\s*$" % name, line.rstrip())
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return out


def main():
    tag, d = sys.argv[1], sys.argv[2]
    fetch = counters("%s/%s_fetch.txt" % (d, tag), "FETCH_SIZE")
    write = counters("%s/%s_write.txt" % (d, tag), "WRITE_SIZE")
    bench = json.load(open("%s/%s_bench_n1.json" % (d, tag)))
    # C2: lane-owned loads (arena below the cooperative switch); the kernel has a third template argument since round 3
    sha = next((k for k in fetch if k.startswith("void mi::sha256_items_kernel<0, false")), "void mi::sha256_items_kernel<0, false>")
    gear = "mi::gear_cdc_small_fast_kernel" if "mi::gear_cdc_small_fast_kernel" in fetch else "mi::gear_cdc_small_kernel"
    synth_kib = write.get("mi::synth_fill_kernel", (0, 0.0))[1]
    bytes_in = bench["config"]["bytes_per_gpu"]
    n_chunks = bench["config"]["chunks_last_batch"]
    out = {
        "round": int(tag.lstrip("r")),
        "source": "profiles/%s_pmc_fetch_size.txt + %s_pmc_write_size.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                  "--kernel-trace -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-with-rows --no-commit-e2e), same "
                  "tools/round_profiles.sh run as the round's bench line" % (tag, tag),
        "kernel": "mi::sha256_items_kernel<0, false> (chunk pass), C2 batch",
        "FETCH_SIZE_KiB_raw": fetch[sha][1],
        "WRITE_SIZE_KiB_raw": write[sha][1],
        "correction": "gfx950: FETCH_SIZE tallies 128-B read requests at 64 B, so reads are doubled "
                      "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE calibrated on synth_fill_kernel "
                      "(%.0f KiB reported for %d bytes written per batch)" % (synth_kib, bytes_in),
        "sha256_items_kernel_bytes_per_launch": int(2 * fetch[sha][1] * 1024 + write[sha][1] * 1024),
        "algorithmic_bytes_per_launch": int(bytes_in + 52 * n_chunks),
        "gear_cdc_small_fast_kernel_bytes_per_launch": int(2 * fetch[gear][1] * 1024 + write[gear][1] * 1024),
        "gear_algorithmic_bytes_per_launch": int(bytes_in + 4 * n_chunks),
    }
    out["sha_traffic_ratio"] = round(out["sha256_items_kernel_bytes_per_launch"] / out["algorithmic_bytes_per_launch"], 3)
    out["gear_traffic_ratio"] = round(out["gear_cdc_small_fast_kernel_bytes_per_launch"] / out["gear_algorithmic_bytes_per_launch"], 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
