"""Ad-hoc timing of the device pipeline (not the driver's bench: see bench.py)."""
import argparse
import json
import os
import sys

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
    a = ap.parse_args()
    with makisu_amd.Engine(flags=a.flags, max_size=a.max_size, min_size=a.min_size, mask_bits=a.mask_bits) as e:
        print(json.dumps(e.device_info()))
        with e.batch() as b:
            b.add_synthetic([a.size] * a.files, None)
            b.run()
            print("first", json.dumps(e.stats()))
            for _ in range(a.steps):
                b.rerun()
                st = e.stats()
                gib = st["bytes_in"] / 2**30
                print("step total %.3f ms  %.1f GiB/s | cdc %.3f sort %.3f sha %.3f (%.1f GB/s) roots %.3f dedup %.3f | chunks %d uniq %d"
                      % (st["ms_total"], gib / st["ms_total"] * 1e3, st["ms_cdc"], st["ms_sort"],
                         st["ms_sha_chunks"], st["bytes_in"] / st["ms_sha_chunks"] / 1e6,
                         st["ms_sha_files"], st["ms_dedup"], st["n_chunks"], st["n_unique"]))


if __name__ == "__main__":
    main()
