#!/bin/bash
# Per dispatch of k_pair (four segments per step): duration beside address-translation and memory-side counters -- the third and fourth segment of a
# buffer are the slow ones on a "fast" box, all four on a "slow" one: what differs?
export TMPDIR=/tmp
REPO=$PWD
O=$REPO/gpurun_out/r04q; mkdir -p $O
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e --check sample --check-lines 0 --no-per-call --no-packed --no-cli --no-multi --no-fastq"
cd /tmp
rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $O/p1 -- python3 $REPO/bench.py $ARGS > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum --output-format csv -d $O/p2 -- python3 $REPO/bench.py $ARGS > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $O/p3 -- python3 $REPO/bench.py $ARGS > $O/p3.log 2>&1
cd $REPO
python3 - $O <<'PY'
import csv, glob, sys, os
from collections import defaultdict
O = sys.argv[1]
for p in ("p1", "p2", "p3"):
    dur = {}
    for f in glob.glob(os.path.join(O, p, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_pair<" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    vals = defaultdict(dict)
    for f in glob.glob(os.path.join(O, p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Dispatch_Id"] in dur:
                vals[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    names = sorted({c for v in vals.values() for c in v})
    print("==", p, "per k_pair dispatch: us |", " | ".join(names))
    for d in sorted(dur, key=lambda x: int(x)):
        print("   %7.1f us | " % dur[d] + " | ".join("%.4g" % vals[d].get(c, float("nan")) for c in names))
PY
find $O -name "*.csv" -size +2M -delete
