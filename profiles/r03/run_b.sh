#!/bin/bash
# Round 3, GPU call b: PMC counters of k_pair next to k_stream on the headline workload (100 M reads, --best).
set -u
O=$PWD/gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
B="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0"
cd /tmp
for k in pair stream; do
  export SEEQ_FUSED_KERNEL=$k
  [ $k = pair ] && unset SEEQ_FUSED_KERNEL
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$k -- python3 $REPO/bench.py $B > $O/trace_$k.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/pmc1_$k -- python3 $REPO/bench.py $B > $O/pmc1_$k.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_SMEM --output-format csv -d $O/pmc2_$k -- python3 $REPO/bench.py $B > $O/pmc2_$k.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
O = "gpurun_out/r03b"
for k in ("pair", "stream"):
    for f in glob.glob("%s/trace_%s/**/*kernel_stats.csv" % (O, k), recursive=True):
        for r in list(csv.DictReader(open(f)))[:8]:
            print(k, "%-50s calls %5s avg_us %10.2f pct %s" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    agg = defaultdict(lambda: defaultdict(list))
    for sub in ("pmc1_", "pmc2_"):
        for f in glob.glob("%s/%s%s/**/*counter_collection.csv" % (O, sub, k), recursive=True):
            for r in csv.DictReader(open(f)):
                kn = r["Kernel_Name"]
                if "k_pair<" in kn or "k_stream<" in kn or "k_exact1<" in kn:
                    agg[kn[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn in agg:
        print(k, kn)
        for c in sorted(agg[kn]):
            v = agg[kn][c]
            print("     %-26s n=%-4d mean=%.5g" % (c, len(v), sum(v) / len(v)))
PY
find $O -name "*.csv" -size +4M -delete
