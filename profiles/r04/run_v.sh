#!/bin/bash
# kernel stats of the packed section (quad walk)
export TMPDIR=/tmp
REPO=$PWD
O=$REPO/gpurun_out/r04v; mkdir -p $O
B="--steps 6 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-multi --no-fastq --check sample --check-lines 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $REPO/bench.py $B > $O/t.log 2>&1
cd $REPO
python3 - $O/t <<'PY'
import csv,glob,sys,os
for f in glob.glob(os.path.join(sys.argv[1],"**","*kernel_stats.csv"),recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]:
        print("  %-70s %6s calls  avg %9.1f us  %5.2f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
find $O -name "*.csv" -size +2M -delete
