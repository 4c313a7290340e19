#!/bin/bash
# What differs between k_pair launches over fast and over slow buffers of ONE process?  bench.py's placement candidates give both kinds;
# per dispatch: duration beside memory-side counters (separate passes), grouped by speed.
export TMPDIR=/tmp
REPO=$PWD
O=$REPO/gpurun_out/r04ai; mkdir -p $O
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --check sample --check-lines 0 --no-per-call --no-packed --no-cli --no-multi --no-fastq"
cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUBBLE_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MISSFIFO_FULL_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TA_BUSY_sum TD_BUSY_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $REPO/bench.py $ARGS > $O/p$i.log 2>&1 || echo "pass $i failed"
done
cd $REPO
python3 - $O <<'PY'
import csv, glob, sys, os
from collections import defaultdict
O = sys.argv[1]
for p in sorted(glob.glob(os.path.join(O, "p[0-9]"))):
    dur = {}
    for f in glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_pair<" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    vals = defaultdict(dict)
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Dispatch_Id"] in dur:
                vals[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    names = sorted({c for v in vals.values() for c in v})
    groups = {"fast (< 0.80 ms)": [], "medium": [], "slow (> 0.895 ms)": []}
    for d, us in dur.items():
        if us < 700: continue                       # the short last segment of a buffer
        g = "fast (< 0.80 ms)" if us < 800 else "slow (> 0.895 ms)" if us > 895 else "medium"
        groups[g].append(d)
    print("== pass", os.path.basename(p), ":", " | ".join(names))
    for g, ds in groups.items():
        if not ds: continue
        print("   %-18s n=%-3d mean us %.1f | " % (g, len(ds), sum(dur[d] for d in ds) / len(ds)) +
              " | ".join("%.4g" % (sum(vals[d].get(c, 0.0) for d in ds) / len(ds)) for c in names))
PY
find $O -name "*.csv" -size +1M -delete
