#!/bin/bash
set -u
O=gpurun_out/r03ax; mkdir -p $O
for g in 16 32 64 16 32 64 8; do
  SEEQ_HIT_GRID=$g timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --check-lines 0 > $O/b_$g.json 2> $O/b_$g.err || break
  python3 -c "
import json; d=json.load(open('$O/b_$g.json')); print('grid $g', round(d['ms_per_step'],3), d['device_ms_per_step'], round(d['roofline']['avg_launch_ms'],4))"
done
