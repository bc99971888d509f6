"""The content-aware commit on a REAL tree: this image's own /usr (15 GB, 114 000 entries: shared libraries of hundreds of megabytes, tens
of thousands of small Python files, symlinks with absolute targets, hard links) as the root file system of a build -- a MemFS rooted at "/"
with every other top-level directory blacklisted.  Three handles: with a ctx, with a ctx and MI_MEMFS_TRUST_CTIME, without a ctx; each commits
the tree (all new), then again (nothing changed).  Checks: the three layer tars have the same TarDigest; the chunk roots of a random sample
of files are the oracle's.  Prints the commit table.        usage: real_tree_commit.py [top = /usr] [sample = 300]
MI_REAL_N_CTXS=k adds a fourth handle that commits over k ctxs on the one device (mi_memfs_commit_layer_n); MI_REAL_WARM=1 reads the tree
once before the handles (the first read of a fresh box pages its image in from remote storage: minutes that are not the engine's)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import makisu_amd as M  # noqa: E402
from commit_cases import oracle_root  # noqa: E402


def main():
    top = sys.argv[1] if len(sys.argv) > 1 else "/usr"
    n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    from oracle import mi_oracle as O
    O.build()
    first = "/" + top.strip("/").split("/")[0]
    blacklist = ["/" + n for n in os.listdir("/") if "/" + n != first]
    if first != top:                                               # a subtree: its siblings on the way down are blacklisted too
        cur = first
        for part in top.strip("/").split("/")[1:]:
            blacklist += [os.path.join(cur, n) for n in os.listdir(cur) if n != part]
            cur = os.path.join(cur, part)
    rows = []
    digests = {}
    all_roots = {}
    if os.environ.get("MI_REAL_WARM") == "1":
        t0, nb = time.perf_counter(), 0
        for dp, dns, fns in os.walk(top):
            for fn in fns:
                p = os.path.join(dp, fn)
                if os.path.isfile(p) and not os.path.islink(p):
                    with open(p, "rb") as fh:
                        while True:
                            b = fh.read(8 << 20)
                            if not b:
                                break
                            nb += len(b)
        print("# warm-up read of %s: %.2f GB in %.1f s" % (top, nb / 1e9, time.perf_counter() - t0))
    k = int(os.environ.get("MI_REAL_N_CTXS", "0"))
    gz = int(os.environ["MI_REAL_GZIP"]) if os.environ.get("MI_REAL_GZIP") else M.GZIP_OFF     # -1: tario's default level; unset: the gzip leg off
    more = [M.Engine(device=0, n_streams=4) for _ in range(max(0, k - 1))]
    with M.Engine(device=0) as eng:
        handles = [("gpu", {"engine": eng}, False), ("gpu_trust_ctime", {"engine": eng}, True), ("cpu_header_only", {}, False)]
        if k > 1:
            handles.append(("gpu_%d_ctxs" % k, {"engine": [eng] + more}, False))
        for name, kw, trust in handles:
            with M.MemFS("/", blacklist=blacklist) as fs:
                if trust:
                    fs.set_options(trust_ctime=True)
                for what in ("all new", "nothing changed"):
                    t0 = time.perf_counter()
                    res = fs.commit_layer(must_scan=True, gzip_level=gz, **kw)
                    dt = time.perf_counter() - t0
                    st = res["stats"]
                    rows.append((name, what, dt, st, res))
                    if what == "all new":
                        digests[name] = str(res["tar_digest"])
                        all_roots[name] = {e["relpath"]: e["root"] for e in res["layer"] if e["kind"] == M.KIND_FILE and "root" in e}
                        if name == "gpu":
                            files = [e for e in res["layer"] if e["kind"] == M.KIND_FILE and "root" in e and e["size"] <= (64 << 20)]
                            rng = np.random.default_rng(5)
                            pick = [files[int(i)] for i in rng.choice(len(files), size=min(n_sample, len(files)), replace=False)]
                            bad = [e["relpath"] for e in pick if oracle_root(O, open("/" + e["relpath"], "rb").read()) != e["root"]]
                            n_links = sum(1 for e in res["layer"] if e["kind"] == M.KIND_SYMLINK)
                            print("%d entries, %d regular files (%.2f GB), %d symlinks, largest file %.0f MB; %d sampled roots against the oracle: %s" %
                                  (res["n_entries"], st["n_layer_files"], st["layer_file_bytes"] / 1e9, n_links,
                                   max(e["size"] for e in res["layer"]) / 1e6, len(pick), "all equal" if not bad else "DIFFER: %s" % bad[:5]))
                            assert not bad
    for name, what, dt, st, res in rows:
        print("%-16s %-16s %7.3f s = walk%s %.3f + diff %.3f + tar %.3f; scan %.3f%s | layer %d entries, %d files | read %d files, %.2f GB (%d trusted)" %
              (name, what, dt, "+stage" if name.startswith("gpu") else "", st["s_walk_stage"], st["s_diff"], st["s_write"], st["s_scan"],
               " (beside)" if st["pipelined"] else "", res["n_entries"], st["n_layer_files"], st["files_opened"], st["file_bytes_read"] / 1e9,
               st["n_content_trusted"]))
    if gz != M.GZIP_OFF:
        for name, what, dt, st, res in rows:
            if what == "all new":
                print("%-16s gzip level %d (MI_GZIP_PROBE=%s): %.2f GB tar -> %.2f GB blob (%.3f), %s" %
                      (name, gz, os.environ.get("MI_GZIP_PROBE", "1"), res["tar_bytes"] / 1e9, res["gzip_bytes"] / 1e9, res["gzip_bytes"] / res["tar_bytes"], res["gzip_digest"][:26]))
    for name, what, dt, st, res in rows:
        if what == "all new" and name.startswith("gpu"):
            print("%-16s verified %d files, %.2f GB, %d chunks fetched twice; arena %.2f GB in %d pieces, moved %d times; %d ctx(s), bytes per ctx %.2f - %.2f GB" %
                  (name, st["n_verified_files"], st["verified_bytes"] / 1e9, st["n_refetched"], st["arena_bytes"] / 1e9, st["arena_pieces"], st["arena_moves"],
                   st["n_ctxs"], st["ctx_bytes_min"] / 1e9, st["ctx_bytes_max"] / 1e9))
    # every root of every handle that scanned: the one-ctx commit's -- among them the files of 256 MiB and more, which the commit over
    # several ctxs SPLITS into parts (their root is computed from the parts' digests) -- and those against the oracle as well
    gpu_names = [n for n in all_roots if all_roots[n]]
    for n in gpu_names[1:]:
        diff = [r for r in all_roots[gpu_names[0]] if all_roots[n].get(r) != all_roots[gpu_names[0]][r]]
        print("%-16s %d roots, %d differ from %s's%s" % (n, len(all_roots[n]), len(diff), gpu_names[0], (": %s" % diff[:5]) if diff else ""))
        assert not diff and len(all_roots[n]) == len(all_roots[gpu_names[0]])
    for name, what, dt, st, res in rows:
        if what == "all new" and st["n_ctxs"] > 1:
            big = [e for e in res["layer"] if e["kind"] == M.KIND_FILE and e["size"] >= (256 << 20)]
            bad = [e["relpath"] for e in big if oracle_root(O, open("/" + e["relpath"], "rb").read()) != e["root"]]
            print("%-16s %d files split over the ctxs as parts (%d of 256 MiB and more, %.2f GB): their roots against the oracle's of the whole files: %s" %
                  (name, st["n_split_files"], len(big), sum(e["size"] for e in big) / 1e9, "all equal" if not bad else "DIFFER: %s" % bad))
            assert not bad and st["n_split_files"] == len(big)
    for e in more:
        e.close()
    same = len(set(digests.values())) == 1
    print("TarDigest %s%s" % (digests["gpu"][:26], " -- the same from all %d handles" % len(digests) if same else " -- DIFFERENT: %s" % digests))
    assert same


if __name__ == "__main__":
    main()
