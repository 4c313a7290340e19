#!/bin/bash
set -u
O=gpurun_out/r03aw; mkdir -p $O
for k in pair stream pair stream; do
  SEEQ_FUSED_KERNEL=$k timeout -k 10 300 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --check-lines 0 > $O/cfg5_$k.json 2> $O/cfg5_$k.err || break
  python3 -c "
import json; d=json.load(open('$O/cfg5_$k.json')); print('$k', round(d['ms_per_step'],3), d['device_ms_per_step'], d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],4))"
done
bash profiles/r03_final_evidence.sh pmc
