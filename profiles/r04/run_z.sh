#!/bin/bash
# placement: does the SIZE of the allocation decide which pages a buffer gets (a 16 / 32 / 64 GiB block instead of 15.1 GB)?
O=gpurun_out/r04z; mkdir -p $O
B="--no-per-call --no-packed --no-cli --no-multi --no-fastq --no-cpu-baseline --no-e2e --check sample --check-lines 0"
SEEQ_BENCH_CAND_BYTES=17179869184,34359738368,68719476736,17179869184 python bench.py $B --placement-candidates 5 > $O/p1.json 2> $O/p1.err || exit 1
python3 - $O/p1.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for i,r in enumerate(d['placement']['candidates']): print(i, r['first_round']['launch_ms'], r['launch_ms'])
print('chosen', d['placement']['chosen'], 'step ms', round(d['ms_per_step'],3), d['per_step']['scan_launch_ms_all'][:4])
PY
