#!/bin/bash
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/r04k
mkdir -p $OUT
bash profiles/r04/box_probe.sh
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --check sample --check-lines 0 --no-per-call --no-cli --no-multi > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print("%-90s calls=%s avg_us=%.1f total_ms=%.3f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
find $OUT -name "*.csv" -size +4M -delete
