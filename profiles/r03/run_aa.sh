#!/bin/bash
# Round 3, GPU call aa: kernel trace of the post-pass (count / best, windows on / off) + PMC of k_exact1 COUNT.
set -u
O=$PWD/gpurun_out/r03aa; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for mode in count best; do
for w in window whole; do
  [ $w = whole ] && export SEEQ_NO_WINDOW=1 || unset SEEQ_NO_WINDOW
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${mode}_$w -- python3 $REPO/profiles/time_scan.py x 100000000 3 $mode > $O/trace_${mode}_$w.log 2>&1
done; done
unset SEEQ_NO_WINDOW
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_best -- python3 $REPO/profiles/time_scan.py x 100000000 3 best > $O/pmc_best.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob
from collections import defaultdict
O = "gpurun_out/r03aa"
for mode in ("count", "best"):
    for w in ("window", "whole"):
        for f in glob.glob("%s/trace_%s_%s/**/*kernel_stats.csv" % (O, mode, w), recursive=True):
            for r in list(csv.DictReader(open(f)))[1:9]:
                print(mode, w, "%-56s calls %5s avg_us %10.2f" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3))
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob("%s/pmc_best/**/*counter_collection.csv" % O, recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_exact1<" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:32]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn in agg:
    print(kn, {c: "%.4g" % (sum(v) / len(v)) for c, v in agg[kn].items()})
PY
find $O -name "*.csv" -size +4M -delete
