#!/bin/bash
# placement candidates: how far apart are N buffers of one process?
O=gpurun_out/r04y; mkdir -p $O
B="--no-per-call --no-packed --no-cli --no-multi --no-fastq --no-cpu-baseline --no-e2e --check sample --check-lines 0"
for i in 1 2; do
python bench.py $B --placement-candidates 5 > $O/p$i.json 2> $O/p$i.err || exit 1
python3 - $O/p$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for i,r in enumerate(d['placement']['candidates']): print(i, r)
print('chosen', d['placement']['chosen'], 'step ms', round(d['ms_per_step'],3), d['per_step']['scan_launch_ms_all'][:4])
PY
done
