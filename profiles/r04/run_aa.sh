#!/bin/bash
# the default bench line with placement candidates (3), then cfg5 and count
O=gpurun_out/r04aa; mkdir -p $O
timeout -k 10 400 python bench.py > $O/best.json 2> $O/best.err; echo "best exit $?"
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq"
timeout -k 10 300 python bench.py --workload cfg5 --steps 20 --warmup 3 $B > $O/cfg5.json 2> $O/cfg5.err; echo "cfg5 exit $?"
timeout -k 10 300 python bench.py --workload all $B > $O/all.json 2> $O/all.err; echo "all exit $?"
python3 - <<'PY'
import json
for w in ("best","cfg5","all"):
    d=json.load(open("gpurun_out/r04aa/%s.json"%w))
    print(w, round(d["value"]/1e9,2), "G lines/s", round(d["ms_per_step"],3), "ms", d["roofline"]["avg_launch_ms"], round(d["roofline"]["frac"],3), round(d["roofline"]["whole_step_frac"],3), (d["results"].get("oracle_check") or {}).get("result"), (d["results"].get("oracle_check") or {}).get("reference_lines_checked"))
    for i,r in enumerate(d["placement"]["candidates"]): print("   cand", i, r["allocated_bytes"], r["launch_ms"])
    print("   chosen", d["placement"]["chosen"])
PY
tail -3 $O/best.err
