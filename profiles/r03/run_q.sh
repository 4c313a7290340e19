#!/bin/bash
# Round 3, GPU call q: do the post-pass kernels really run under the next segment's k_pair?  (kernel trace timestamps)
set -u
O=$PWD/gpurun_out/r03q; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $REPO/profiles/time_scan.py overlap 100000000 2 best > $O/trace.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/r03q/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40 kernels
t0 = int(rows[-60]["Start_Timestamp"])
for r in rows[-60:]:
    print("%-40s q=%s start %9.1f us  dur %8.1f us" % (r["Kernel_Name"][:40], r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
