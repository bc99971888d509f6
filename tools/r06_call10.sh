#!/bin/bash
# round 6, GPU call 10: the commit on a real root file system (this image's /usr, 15 GB) with the round's code + over 2 ctxs; the commit
# table at a million files; the commit of 48 x 128 MiB over 1, 2 and 4 ctxs on the one device
mkdir -p gpurun_out/big
MI_REAL_WARM=1 MI_REAL_N_CTXS=2 timeout 2400 python tools/real_tree_commit.py /usr 300 > gpurun_out/big/r06_real_tree_commit.txt 2>&1
tail -14 gpurun_out/big/r06_real_tree_commit.txt | cut -c1-260
timeout 600 python tools/commit_layer_bench.py 1000000 4096 > gpurun_out/big/r06_commit_e2e_1m.txt 2>&1
grep -E "all new|nothing|rewritten|^    (gpu|cpu)" gpurun_out/big/r06_commit_e2e_1m.txt | cut -c1-200
python - > gpurun_out/big/r06_commit_n_ctxs.txt 2>&1 <<'PY'
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import makisu_amd as M
from commit_layer_bench import _make_tree
root = tempfile.mkdtemp(prefix="mi_nctx_", dir="/dev/shm")
try:
    _make_tree(root, 48, 128 << 20, 8)
    engines = [M.Engine(device=0) for _ in range(4)]
    with M.MemFS(root) as warm:
        warm.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF, engine=engines, want_layer=False)      # every ctx's first use
    print("# 48 x 128 MiB all new / nothing changed, gzip off, k ctxs on ONE device (one PCIe link: no speed-up to expect -- the claim is 'no slower')")
    for k in (1, 2, 4, 1, 2, 4):
        with M.MemFS(root) as fs:
            eng = engines[0] if k == 1 else engines[:k]
            row = []
            for what in ("all new", "nothing changed"):
                t0 = time.perf_counter()
                res = fs.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF, engine=eng, want_layer=False)
                st = res["stats"]
                row.append("%s %.3f s (walk+stage %.3f, scan %.3f, tar %.3f)" % (what, time.perf_counter() - t0, st["s_walk_stage"], st["s_scan"], st["s_write"]))
            print("k = %d: %s | %s | per ctx %.2f - %.2f GB, verified %d files" % (k, row[0], row[1], st["ctx_bytes_min"] / 1e9, st["ctx_bytes_max"] / 1e9, st["n_verified_files"]))
    for e in engines:
        e.close()
finally:
    shutil.rmtree(root, ignore_errors=True)
PY
cat gpurun_out/big/r06_commit_n_ctxs.txt
