#!/bin/bash
set -u
O=gpurun_out/r03bg; mkdir -p $O
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace -- python3 $REPO/profiles/chrom_sweep.py --cells 20:5,42:14 --no-ref > $REPO/$O/sweep.jsonl 2> $REPO/$O/sweep.err
cd $REPO
tail -4 $O/sweep.err
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r03bg/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-64s calls %5s avg_us %12.2f total_ms %9.2f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $O -name "*.csv" -size +2M -delete
