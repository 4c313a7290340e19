#!/bin/bash
set -u
O=gpurun_out/r03bb; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_pattern" > $O/t.log 2>&1; rc=$?; echo "tests exit $rc"; tail -5 $O/t.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 800 python3 profiles/multi_bench.py 10000000 > $O/multi.jsonl 2> $O/multi.err; echo "exit $?"; tail -3 $O/multi.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r03bb/multi.jsonl"):
    d = json.loads(l); print(d["set"], "| records:", d["best_records"]["one_pass_ms"], "vs", d["best_records"]["sequential_ms"], "x", d["best_records"]["speedup"], "| count:", d["count_lines"]["one_pass_ms"], "vs", d["count_lines"]["sequential_ms"], "x", d["count_lines"]["speedup"])
PY
bash profiles/r03/run_ap.sh 2>&1 | tail -38
