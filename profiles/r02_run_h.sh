#!/bin/bash
# Round 2, GPU call h: k_stream2 (lane stretches of 1 KB) -- GPU tests, A/B against k_stream on one box, kernel trace.
set -u
O=gpurun_out/r02h; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call"
timeout -k 10 200 python bench.py $B > $O/bench_best.json 2> $O/bench_best.err; echo "best exit $?"
SEEQ_STREAM_V1=1 timeout -k 10 200 python bench.py $B --check-lines 0 > $O/bench_best_v1.json 2> $O/bench_best_v1.err; echo "best v1 exit $?"
SEEQ_STREAM_LAZY=1 timeout -k 10 200 python bench.py $B --check-lines 0 > $O/bench_best_lazy.json 2> $O/bench_best_lazy.err; echo "best lazy exit $?"
REPO=$PWD; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_best -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-e2e --no-per-call --no-cpu-baseline --check-lines 0 > $REPO/$O/prof_best.log 2>&1
cd $REPO
find $O -name "*.csv" -size +8M -delete
for f in $O/bench_*.json; do echo "== $f"; python3 - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d[k] for k in ("value","ms_per_step")}, d["device_ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["results"].get("oracle_check"))
except Exception as e: print("ERR",e)
PY
done
tail -n 3 $O/*.err
head -14 $O/prof_best/*/*_kernel_stats.csv | cut -c1-200
