"""Ad-hoc timing of the device pipeline (not the driver's bench: see bench.py)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=100000)
    ap.add_argument("--size", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--max-size", type=int, default=65536)
    ap.add_argument("--min-size", type=int, default=2048)
    ap.add_argument("--mask-bits", type=int, default=13)
    ap.add_argument("--inflight", type=int, default=1, help="batches in flight (1 = serial steps)")
    ap.add_argument("--lib", default=None, help="dev: load this build of the library instead")
    a = ap.parse_args()
    if a.lib:
        makisu_amd._build.LIB = os.path.abspath(a.lib)
        makisu_amd._build.needs_build = lambda: False
    with makisu_amd.Engine(flags=a.flags, max_size=a.max_size, min_size=a.min_size,
                           mask_bits=a.mask_bits) as e:
        print(json.dumps(e.device_info()))
        batches = []
        for i in range(a.inflight):
            b = e.batch()
            b.add_synthetic([a.size] * a.files, [i * a.files + j for j in range(a.files)])
            b.run()
            batches.append(b)
        print("first", json.dumps(e.stats()))
        for rep in range(3):
            t0 = time.perf_counter()
            pending = []
            for k in range(a.steps):
                b = batches[k % a.inflight]
                if len(pending) == a.inflight:
                    pending.pop(0).wait()
                b.submit()
                pending.append(b)
            while pending:
                pending.pop(0).wait()
            dt = (time.perf_counter() - t0) / a.steps
            st = e.stats()
            gib = st["bytes_in"] / 2**30
            print("inflight %d: %.3f ms/step  %.1f GiB/s | last step: total %.3f cdc %.3f sort %.3f sha %.3f (%.1f GB/s) roots %.3f dedup %.3f | chunks %d uniq %d"
                  % (a.inflight, dt * 1e3, gib / dt, st["ms_total"], st["ms_cdc"], st["ms_sort"],
                     st["ms_sha_chunks"], st["bytes_in"] / st["ms_sha_chunks"] / 1e6,
                     st["ms_sha_files"], st["ms_dedup"], st["n_chunks"], st["n_unique"]))
        for b in batches:
            b.free()


if __name__ == "__main__":
    main()
