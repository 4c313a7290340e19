#!/bin/bash
# kernel stats of the --all and cfg5 workloads (k_emit_all in place)
export TMPDIR=/tmp
REPO=$PWD
O=$REPO/gpurun_out/r04t; mkdir -p $O
B="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq --check sample --check-lines 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/all -- python3 $REPO/bench.py --workload all $B > $O/all.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5 -- python3 $REPO/bench.py --workload cfg5 $B > $O/cfg5.log 2>&1
cd $REPO
for w in all cfg5; do
echo "== $w"
python3 - $O/$w <<'PY'
import csv,glob,sys,os
for f in glob.glob(os.path.join(sys.argv[1],"**","*kernel_stats.csv"),recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("  %-60s %6s calls  avg %9.1f us  %5.2f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
done
find $O -name "*.csv" -size +2M -delete
