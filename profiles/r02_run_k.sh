#!/bin/bash
# Round 2, GPU call k: two-chain k_stream2 (opt-in) parity + A/B; pipe streaming test; whole GPU suite.
set -u
O=gpurun_out/r02k; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "pipe or boundaries or edge_buffers or batch_scan" > $O/pytest_sel.log 2>&1; echo "pytest sel exit $?" >> $O/pytest_sel.log
tail -12 $O/pytest_sel.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call"
SEEQ_STREAM_V2=1 timeout -k 10 200 python bench.py $B > $O/bench_best_v2.json 2> $O/bench_best_v2.err; echo "best v2 exit $?"
SEEQ_STREAM_V2=1 SEEQ_STREAM_LAZY=1 timeout -k 10 200 python bench.py $B --check-lines 0 > $O/bench_best_v2lazy.json 2> $O/bench_best_v2lazy.err; echo "best v2 lazy exit $?"
timeout -k 10 200 python bench.py $B --check-lines 0 > $O/bench_best_v1.json 2> $O/bench_best_v1.err; echo "best v1 exit $?"
for f in $O/bench_*.json; do echo "== $f"; python3 - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d[k] for k in ("value","ms_per_step")}, d["device_ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["results"].get("oracle_check"))
except Exception as e: print("ERR",e)
PY
done
tail -n 3 $O/*.err
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
