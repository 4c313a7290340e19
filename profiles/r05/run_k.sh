#!/bin/bash
# round 5, call K: where do 2 ms of the sweep's cell m = 42, k = 9 go? (kernel stats)
out=$PWD/gpurun_out/r05_k; mkdir -p $out
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
for cell in 42:9 42:12; do
tag=$(echo $cell | tr ':' '_')
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$tag -- python3 $REPO/profiles/chrom_sweep.py --cells $cell --no-ref > $out/t_$tag.log 2>&1
python3 - $out/t_$tag <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        if "at::native" in r["Name"] or "rocclr" in r["Name"]: continue
        print("   %-90s calls %s avg_us %.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find $out -name "*.csv" -size +1M -delete
