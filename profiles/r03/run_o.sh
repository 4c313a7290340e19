#!/bin/bash
# Round 3, GPU call o: PMC of k_verify next to k_exact1 COUNT.
set -u
O=$PWD/gpurun_out/r03o; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for k in verify exact1; do
  [ $k = exact1 ] && export SEEQ_NO_VERIFY=1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_$k -- python3 $REPO/profiles/time_scan.py $k 100000000 3 best > $O/pmc_$k.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2_$k -- python3 $REPO/profiles/time_scan.py $k 100000000 3 best > $O/pmc2_$k.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob
from collections import defaultdict
O = "gpurun_out/r03o"
for k in ("verify", "exact1"):
    agg = defaultdict(lambda: defaultdict(list))
    for sub in ("pmc_", "pmc2_"):
        for f in glob.glob("%s/%s%s/**/*counter_collection.csv" % (O, sub, k), recursive=True):
            for r in csv.DictReader(open(f)):
                kn = r["Kernel_Name"]
                if "k_verify<" in kn or "k_exact1<1" in kn:
                    agg[kn[:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn in agg:
        print(k, kn)
        for c in sorted(agg[kn]):
            v = agg[kn][c]
            print("     %-26s n=%-4d mean=%.5g" % (c, len(v), sum(v) / len(v)))
PY
find $O -name "*.csv" -size +4M -delete
