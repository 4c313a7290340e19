#!/bin/bash
# round 5, call AU: the text buffer chosen with the run's OWN scan context (seeqdevTextAllocFor) -- does the probe's ranking now carry over to the timed steps?
out=$PWD/gpurun_out/r05_au; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python3 profiles/r05/workspace_probe.py 2>&1 | grep -v amdgpu | tee $out/workspace_probe.txt
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --sections none --check-lines 0 --first-steps 10 --steps 10 --warmup 2 > $out/b_$i.json 2> $out/b_$i.err || { echo "bench failed"; tail -3 $out/b_$i.err; exit 1; }
  python3 - $out/b_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
pr=d["placement"]["probe_forward_ms"]
print("value %.2f G lines/s  step %.3f ms  chosen-buffer launch %.4f  probe of the chosen %.3f (/4 = %.4f)  probes min %.3f max %.3f  first allocation %.2f G lines/s launch %s" % (d["value"]/1e9, d["ms_per_step"], d["roofline"]["avg_launch_ms"], pr[d["placement"]["chosen"]], pr[d["placement"]["chosen"]]/4, min(pr), max(pr), d["first_allocation"]["value"]/1e9, d["first_allocation"]["scan_launch_ms"]))
PY
done
