#!/bin/bash
# k_pair over the quad table (SEEQ_PAIR_QUAD=1: four text bytes per gather, the two-part filter) against the shipped kernel
set -o pipefail
O=gpurun_out/r04an; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "PAIR_QUAD" > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && { tail -40 $O/pytest.log; exit $rc; }
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq --steps 50"
for v in 1 0 1 0; do
  SEEQ_PAIR_QUAD=$v timeout -k 10 300 python bench.py $B > $O/q$v.json 2> $O/q$v.err; echo "quad=$v exit $?"
  python3 - $O/q$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
chk=d["results"].get("oracle_check") or {}
print("   ", round(d["value"]/1e9,2), "G lines/s", round(d["ms_per_step"],3), "ms; scan", round(d["roofline"]["avg_launch_ms"],4), "post", round(d["device_ms_per_step"]["compaction_exact_records"],3), chk.get("result"), chk.get("reference_lines_checked"), chk.get("reference_result"))
print("      cands", [r["launch_ms"][0] for r in d["placement"]["candidates"]])
PY
done
