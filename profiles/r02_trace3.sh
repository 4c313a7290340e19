#!/bin/bash
# kernel-trace stats only (no counters) of the three record workloads, short runs
set -u
export TMPDIR=/tmp
REPO=$PWD
for w in best all cfg5; do
  OUT=$REPO/gpurun_out/trace3_$w; mkdir -p $OUT
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0 > "$OUT/trace.log" 2>&1)
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  echo "== $w"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-70s calls %5s avg_us %10.2f total_us %10.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
  find $OUT -name "*.csv" -size +2M -delete
done
