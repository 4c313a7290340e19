#!/bin/bash
# round 5, call M: kernel stats of the FASTQ shape under SQ_IGNORE on k_pair (where does the post-pass go?)
out=$PWD/gpurun_out/r05_m; mkdir -p $out
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 $REPO/profiles/fastq_shape_bench.py 25000000 best fastq ignore > $out/t.log 2>&1
python3 - $out/t <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]:
        if "at::native" in r["Name"] or "rocclr" in r["Name"]: continue
        print("   %-90s calls %s avg_us %.1f total_ms %.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $out -name "*.csv" -size +1M -delete
