#!/bin/bash
# round 5, call O: the whole GPU suite (stress variants as background jobs of their module, CLI pools of four) + FASTQ shape + default line
out=$PWD/gpurun_out/r05_o; mkdir -p $out
export TMPDIR=/tmp
s=$(date +%s)
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=25 > $out/pytest.log 2>&1; echo "suite exit $? in $(( $(date +%s) - s )) s"; tail -32 $out/pytest.log
for m in fail convert ignore; do
  timeout -k 10 200 python3 profiles/fastq_shape_bench.py 25000000 best fastq $m > $out/fastq_$m.json 2> $out/fastq_$m.err; echo "$m exit $?"; cut -c1-330 $out/fastq_$m.json
done
timeout -k 10 600 python bench.py > $out/bench_best.json 2> $out/bench_best.err; echo "bench exit $?"; python3 - $out/bench_best.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(round(d["value"]/1e9,3), round(d["ms_per_step"],3), d["device_ms_per_step"], round(d["roofline"]["avg_launch_ms"],4), round(d["roofline"]["frac"],3), round(d["roofline"]["whole_step_frac"],3), (d["results"].get("oracle_check") or {}).get("reference_lines_checked"), "first", round(d["first_allocation"]["value"]/1e9,3), d["first_allocation"]["scan_launch_ms"])
print({m: (round(v["gb_per_s"]), v["kernel"], v.get("identical_to_reference_count")) for m, v in d["fastq_shape"]["modes"].items()})
c=d["cfg5"]; print("cfg5", round(c["value"]/1e9,3), round(c["ms_per_step"],3), c["device_ms_per_step"], round(c["whole_step_frac"],3))
PY
