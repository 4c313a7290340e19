#!/bin/bash
# Round 2, GPU call c: lazy alphabet check (k_stream without it under SQ_FAIL), SQ_CONVERT through the SUB variant.
set -u
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call"
timeout 600 python bench.py $B > $O/bench_best.json 2> $O/bench_best.err; echo "best exit $?"
SEEQ_STREAM_CHECK=1 timeout 600 python bench.py $B --check-lines 0 > $O/bench_best_chk.json 2> $O/bench_best_chk.err; echo "best chk exit $?"
SEEQ_OVERLAP=0 timeout 600 python bench.py $B --check-lines 0 > $O/bench_best_nooverlap.json 2> $O/bench_best_nooverlap.err; echo "no-overlap exit $?"
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-e2e --no-per-call --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 exit $?"
timeout 600 python bench.py --workload count $B > $O/bench_count.json 2> $O/bench_count.err; echo "count exit $?"
for x in fail convert ignore; do timeout 300 python profiles/fastq_shape_bench.py 5000000 best fastq $x > $O/fastq_$x.json 2> $O/fastq_$x.err; echo "fastq $x exit $?"; cat $O/fastq_$x.json; done
REPO=$PWD; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_best -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-e2e --no-per-call --no-cpu-baseline --check-lines 0 > $REPO/$O/prof_best.log 2>&1
cd $REPO
find $O -name "*.csv" -size +8M -delete
for f in $O/bench_*.json; do echo "== $f"; python3 - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d[k] for k in ("value","ms_per_step")}, d["device_ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["results"].get("oracle_check"))
except Exception as e: print("ERR",e)
PY
done
tail -n 3 $O/*.err
head -8 $O/prof_best/*/*_kernel_stats.csv | cut -c1-160
