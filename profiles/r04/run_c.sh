#!/bin/bash
# kernel-trace stats of the default line (100 M reads, --best), old and new exact pass
set -u
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/r04c
mkdir -p $OUT
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-e2e --check-lines 0 --no-per-call --no-packed --no-cli --no-multi"
cd /tmp
for v in new old; do
  if [ $v = old ]; then export SEEQ_VERIFY=old; else unset SEEQ_VERIFY; fi
  for wl in best all; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_${v}_$wl -- python3 $REPO/bench.py $ARGS --workload $wl > $OUT/trace_${v}_$wl.log 2>&1
    f=$(find $OUT/trace_${v}_$wl -name "*kernel_stats.csv" | head -1)
    echo "== $v $wl" >> $OUT/stats.txt
    python3 - "$f" >> $OUT/stats.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-100s calls=%s avg_us=%.1f total_ms=%.3f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
  done
done
find $OUT -name "*.csv" -size +4M -delete
cat $OUT/stats.txt
