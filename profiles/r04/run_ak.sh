#!/bin/bash
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq --check sample --steps 20"
for i in 1 2; do
python bench.py $B 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print('run', round(d['ms_per_step'],3), d['results']['oracle_check']['result'], d['roofline']['traffic'] is not None, [r['launch_ms'][0] for r in d['placement']['candidates']])"
done
