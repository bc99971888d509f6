# round 6, GPU call 8: the whole gpu suite (N-ctx commit, sums, long strings, arenas), the bench line as the driver runs it
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/r06_gputests_nctx.txt
grep -E "passed|failed|FAILED" gpurun_out/r06_gputests_nctx.txt
timeout 900 python bench.py > gpurun_out/r06_bench_n1_candidate.json 2> gpurun_out/r06_bench_n1_candidate.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_n1_candidate.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("with_rows_ratio"))
print(json.dumps(d["cpu_baseline"].get("commit_s")))
print(d.get("commit_e2e", {}).get("first_use_s"))
PY
tail -3 gpurun_out/r06_bench_n1_candidate.err
