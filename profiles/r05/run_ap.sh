#!/bin/bash
# round 5, call AP: kernel stats of the sweep's cell m = 34, k = 7 with the restart table and with the absorbing one
out=$PWD/gpurun_out/r05_ap; mkdir -p $out
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/restart -- python3 $REPO/profiles/chrom_sweep.py --no-ref --cells 34:7 > $out/restart.log 2>&1
SEEQ_NO_WINDOW=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/absorb -- python3 $REPO/profiles/chrom_sweep.py --no-ref --cells 34:7 > $out/absorb.log 2>&1
cd $REPO
python3 - $out <<'PY'
import csv, glob, sys
for v in ("restart", "absorb"):
    print("==", v)
    for f in glob.glob("%s/%s/**/*kernel_stats.csv" % (sys.argv[1], v), recursive=True):
        for r in list(csv.DictReader(open(f)))[:14]:
            if "at::native" in r["Name"] or "rocclr" in r["Name"] or "elementwise" in r["Name"]: continue
            print("   %-90s calls %5s avg_us %9.1f total_ms %8.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $out -name "*.csv" -size +1M -delete
