#!/bin/bash
set -u
O=$PWD/gpurun_out/r03ab; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for mode in count best; do
for w in window whole; do
  [ $w = whole ] && export SEEQ_NO_WINDOW=1 || unset SEEQ_NO_WINDOW
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_${mode}_$w -- python3 $REPO/profiles/time_scan.py x 100000000 2 $mode > $O/pmc_${mode}_$w.log 2>&1
done; done
cd $REPO
python3 - <<'PY'
import csv, glob
from collections import defaultdict
O = "gpurun_out/r03ab"
for mode in ("count", "best"):
    for w in ("window", "whole"):
        agg = defaultdict(lambda: defaultdict(list))
        for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (O, mode, w), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_exact1<1" in r["Kernel_Name"]:
                    agg[r["Kernel_Name"][:32]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kn in agg:
            print(mode, w, kn, {c: "%.4g" % (sum(v) / len(v)) for c, v in agg[kn].items()})
PY
find $O -name "*.csv" -size +4M -delete
