#!/bin/bash
# Round 2, GPU call n: the committed evidence of this round -- bench lines (default, count, all, cfg5), rocprofv3 kernel
# stats + PMC passes for the default workload and cfg5, CLI wall clock.
set -u
O=gpurun_out/r02n; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > $O/bench_best.json 2> $O/bench_best.err; echo "best exit $?"
timeout -k 10 300 python bench.py --workload count --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call > $O/bench_count.json 2> $O/bench_count.err; echo "count exit $?"
timeout -k 10 300 python bench.py --workload all --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call > $O/bench_all.json 2> $O/bench_all.err; echo "all exit $?"
timeout -k 10 600 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-e2e --no-per-call > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 exit $?"
for f in $O/bench_*.json; do echo "== $f"; python3 - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d[k] for k in ("value","ms_per_step")}, d["device_ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["results"].get("oracle_check"), d.get("cpu_baseline"))
except Exception as e: print("ERR",e)
PY
done
tail -n 2 $O/*.err
