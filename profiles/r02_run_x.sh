#!/bin/bash
# Round 2, GPU call x: k_stream without the per-tile scratch reload (64-bit uniform compare) -- tests, bench, FETCH_SIZE.
set -u
O=gpurun_out/r02x; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "not cli_golden" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0"
timeout -k 10 300 python bench.py $B > $O/bench_best.json 2> $O/bench_best.err
python3 -c "
import json; d=json.load(open('$O/bench_best.json')); print('best', round(d['ms_per_step'],3), d['device_ms_per_step'], round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"
REPO=$PWD; cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE"; do
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/$O/pmc_$c -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0 > $REPO/$O/pmc_$c.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv,glob
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("gpurun_out/r02x/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_stream<" in r["Kernel_Name"] and r["Counter_Name"]==c]
        print(c, len(v), sum(v)/max(1,len(v)))
PY
find $O -name "*.csv" -size +2M -delete
