#!/bin/bash
# does the chosen buffer keep its speed once the other candidates are freed (before the warm-up)?
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq --check sample --check-lines 0 --steps 20"
for v in 1 "" 1 "" 1 1; do
SEEQ_BENCH_FREE_SPARES=$v python bench.py $B 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); c=d['placement']['candidates'][d['placement']['chosen']]
print('free_spares=$v', 'chosen', d['placement']['chosen'], 'probe', c['launch_ms'], 'steps', d['per_step']['scan_launch_ms_all'][:4], round(d['ms_per_step'],3))"
done
