#!/bin/bash
O=gpurun_out/r04ao; mkdir -p $O
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq --steps 50"
for v in "1 1" "0 0" "1 1" "0 1"; do
  set -- $v
  SEEQ_PAIR_QUAD=$1 SEEQ_PAIR_PF=$2 timeout -k 10 300 python bench.py $B > $O/q.json 2> $O/q.err; echo "quad=$1 pf=$2 exit $?"
  python3 - $O/q.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
chk=d["results"].get("oracle_check") or {}
print("   ", round(d["value"]/1e9,2), "G lines/s", round(d["ms_per_step"],3), "ms; scan", round(d["roofline"]["avg_launch_ms"],4), "post", round(d["device_ms_per_step"]["compaction_exact_records"],3), chk.get("result"), chk.get("reference_result"))
print("      cands", [r["launch_ms"][0] for r in d["placement"]["candidates"]])
PY
done
