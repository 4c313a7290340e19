#!/bin/bash
# Round 3, GPU call n: kernel trace of the post-pass with k_verify / with k_exact1.
set -u
O=$PWD/gpurun_out/r03n; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_verify -- python3 $REPO/profiles/time_scan.py verify 100000000 4 best > $O/trace_verify.log 2>&1
SEEQ_NO_VERIFY=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_exact1 -- python3 $REPO/profiles/time_scan.py exact1 100000000 4 best > $O/trace_exact1.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob
O = "gpurun_out/r03n"
for k in ("verify", "exact1"):
    for f in glob.glob("%s/trace_%s/**/*kernel_stats.csv" % (O, k), recursive=True):
        for r in list(csv.DictReader(open(f)))[:16]:
            print(k, "%-56s calls %5s avg_us %10.2f" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -2 $O/trace_verify.log $O/trace_exact1.log
find $O -name "*.csv" -size +4M -delete
