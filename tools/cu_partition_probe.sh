#!/bin/bash
# Round 5, first GPU call (about one GPU minute): do two batches in flight do better SHARING THE DEVICE IN SPACE than in
# time?  Today both batches' streams see all 256 CUs: the Gear marking of one (memory-bound, 0.59 of HBM peak, 62 % of its
# time waiting) and the hashing of the other (VALU issue-bound) add up to 5.55 ms per C2 step against 5.3 of pure issue
# time -- they do not hide each other (five probes).  With MI_BATCH_CU_MASKS every batch gets compute units
# of its own: its hashing then takes twice as long on half the CUs, its marking -- IF it is bound by what one CU can have
# in flight rather than by the memory system -- less than twice.  Per 6.55 GB step and batch: marking 1.36 * 2 / k +
# hashing 8.26 + 0.6 ms; two batches side by side.  k = 1 (marking scales with CUs): 1 050 GiB/s, worse than today's
# 1 100; k = 2 (marking does not care): 1 196 GiB/s.  The probe measures k and the step for several ways of cutting the
# device (which mask bit is which CU of which XCD is the runtime's business: halves, interleaved quarters, 64 + 192).
#   tools/cu_partition_probe.sh            (on the GPU box, from the repo root; prints one line per split)
set -u
cd "$(dirname "$0")/.."
MI_RUN_EXPERIMENTS=1 timeout 300 python -m pytest tests/test_gpu_sha_schemes.py -q -x -k compute_units 2>&1 | tail -2
run() {   # label, masks, extra bench args
    local out
    out=$(MI_BATCH_CU_MASKS="$2" timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 2>/dev/null | tail -1)
    python - "$1" "$out" <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    j = json.loads(line)
    r = j.get("roofline", {})
    one = j.get("one_batch_at_a_time") or {}
    ph = j.get("serial_phase_ms") or {}
    print("%-34s value %8.1f GiB/s  step %.3f ms | one batch at a time: %s | sha launch %.3f ms | serial phases %s" %
          (label, j["value"], j["ms_per_step"], one.get("value"), r.get("avg_launch_ms", 0), ph))
except Exception as e:
    print("%-34s no line (%s): %s" % (label, e, line[:200]))
PY
}
run "all CUs, two in flight (today)"  ""                                  ""
run "halves 0-127 | 128-255"          "0-127,128-255"                     ""
run "interleaved 64s"                 "0-63+128-191,64-127+192-255"       ""
run "interleaved 32s"                 "0-31+64-95+128-159+192-223,32-63+96-127+160-191+224-255" ""
run "even / odd groups of 8"          "$(python - <<'PY'
a = "+".join("%d-%d" % (k, k + 7) for k in range(0, 256, 16))
b = "+".join("%d-%d" % (k, k + 7) for k in range(8, 256, 16))
print(a + "," + b)
PY
)" ""
run "three batches, thirds"           "0-84,85-169,170-255"               "--inflight 3"
run "one batch on 128 CUs (k)"        "0-127"                             "--inflight 1"
run "one batch on 64 CUs (k)"         "0-63"                              "--inflight 1"
run "one batch, all CUs"              ""                                  "--inflight 1"
