"""SURVEY.md 8(d) "report table per config", as markdown, from the bench lines under profiles/.
usage: python tools/report_table.py r02"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
rows = [("C2", "n1"), ("C3", "c3"), ("C4, one rank's shard", "c4_one_shard"), ("C5 (Zipf s=1.1)", "c5"),
        ("C5U (2^U(10,30), rounds 1-2)", "c5u")]
print("| config | bytes / step | chunks (last batch) | serial phases: CDC / sort / SHA / roots / marking [ms] | step [ms] | GiB/s (one batch at a time) | path % of HBM | "
      "SHA pass % of HBM / of VALU roof (serial) | CPU 1 thread / N threads [GiB/s] | parity vs oracle |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, key in rows:
    path = os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, key))
    if not os.path.exists(path):
        continue
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    r, c, ph = d["roofline"], d["config"], d.get("serial_phase_ms", {})
    cb = d.get("cpu_baseline")
    cpu = "%.2f / %.1f (N = %d)" % (cb["single_thread_GiBps"], cb["value"], cb["cores"]) if cb else "—"
    chk = d.get("dedup_check") or {}
    parity = "cut points, chunk digests, roots: bit-exact (tests)"
    if chk:
        parity += "; unique count %s closed form%s" % (
            "=" if chk.get("ok") else "≠",
            " (+%d 1-byte tail coincidences)" % chk["short_chunk_coincidences"] if chk.get("short_chunk_coincidences") else "")
    one = d.get("one_batch_at_a_time")
    print("| %s | %.2f GB (%d batch(es) in flight) | %s | %.2f / %.2f / %.2f / %.2f / %.2f | %.2f | %.1f%s | %.1f | %.1f / %.1f | %s | %s |" % (
        name, c["job_bytes_per_step"] / 1e9, c["batches_in_flight"], "{:,}".format(c["chunks_last_batch"]).replace(",", " "),
        ph.get("ms_cdc", 0), ph.get("ms_sort", 0), ph.get("ms_sha_chunks", 0), ph.get("ms_sha_files", 0),
        ph.get("ms_dedup", 0), d["ms_per_step"], d["value"], " (%.1f, %.2f ms)" % (one["value"], one["ms_per_step"]) if one else "",
        100 * r["path_frac"],
        100 * r.get("serial_frac", r["frac"]), 100 * r.get("serial_frac_of_valu_roof", r["frac_of_valu_roof"]), cpu, parity))
