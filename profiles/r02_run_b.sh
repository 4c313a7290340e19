#!/bin/bash
# Round 2, GPU call b: overlapped post-pass A/B, single-launch seeqStringMatch, GPU tests.
set -u
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e"
timeout 600 python bench.py $B > $O/bench_best_overlap.json 2> $O/bench_best_overlap.err; echo "overlap exit $?"
SEEQ_OVERLAP=0 timeout 600 python bench.py $B --no-per-call > $O/bench_best_nooverlap.json 2> $O/bench_best_nooverlap.err; echo "no-overlap exit $?"
SEEQ_OVERLAP=0 SEEQ_DFA_WGS=1 timeout 600 python bench.py $B --no-per-call --check-lines 0 > $O/bench_best_wgs1.json 2> $O/bench_best_wgs1.err; echo "wgs1 exit $?"
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-e2e --no-per-call --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 exit $?"
timeout 600 python bench.py --workload count $B --no-per-call > $O/bench_count.json 2> $O/bench_count.err; echo "count exit $?"
REPO=$PWD; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_best -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-e2e --no-per-call --no-cpu-baseline --check-lines 0 > $REPO/$O/prof_best.log 2>&1
cd $REPO
find $O -name "*.csv" -size +8M -delete
for f in $O/bench_*.json; do echo "== $f"; python3 - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d[k] for k in ("value","ms_per_step")}, d["device_ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("per_call"), d["results"].get("oracle_check"))
except Exception as e: print("ERR",e)
PY
done
tail -3 $O/*.err
head -12 $O/prof_best/*/*_kernel_stats.csv | cut -c1-160
