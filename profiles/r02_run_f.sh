#!/bin/bash
# Round 2, GPU call f: stream_layout microbenchmark (HBM rate of per-lane stretches), lazy check with independent loads.
set -u
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 120 profiles/microbench/stream_layout > $O/stream_layout.txt 2>&1; echo "stream_layout exit $?"; cat $O/stream_layout.txt
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "every_byte or boundaries or fuzz or segments" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0"
timeout -k 10 200 python bench.py $B > $O/bench_best.json 2> $O/bench_best.err; echo "best exit $?"
SEEQ_STREAM_CHECK=1 timeout -k 10 200 python bench.py $B > $O/bench_best_chk.json 2> $O/bench_best_chk.err; echo "best chk exit $?"
REPO=$PWD; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_best -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-e2e --no-per-call --no-cpu-baseline --check-lines 0 > $REPO/$O/prof_best.log 2>&1
cd $REPO
find $O -name "*.csv" -size +8M -delete
for f in $O/bench_*.json; do echo "== $f"; python3 - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d[k] for k in ("value","ms_per_step")}, d["device_ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
except Exception as e: print("ERR",e)
PY
done
head -14 $O/prof_best/*/*_kernel_stats.csv | cut -c1-200
