#!/bin/bash
# k_pair with the next tile prefetched into LDS (SEEQ_PAIR_PF=1) against the shipped kernel: full reference check, placement candidates' launch times
O=gpurun_out/r04ac; mkdir -p $O
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq"
for v in 1 0 1 0; do
  SEEQ_PAIR_PF=$v timeout -k 10 300 python bench.py $B > $O/pf$v.json 2> $O/pf$v.err; echo "pf=$v exit $?"
  python3 - $O/pf$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
chk=d["results"].get("oracle_check") or {}
print("   ", round(d["value"]/1e9,2), "G lines/s", round(d["ms_per_step"],3), "ms; k_pair", round(d["roofline"]["avg_launch_ms"],4), "post", round(d["device_ms_per_step"]["compaction_exact_records"],3), chk.get("result"), chk.get("reference_lines_checked"), chk.get("reference_result"))
for i,r in enumerate(d["placement"]["candidates"]): print("      cand", i, r["launch_ms"])
PY
done
