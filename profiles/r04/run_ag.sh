#!/bin/bash
# placement: the candidates' device addresses beside their speed, several processes on one box
O=gpurun_out/r04ag; mkdir -p $O
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq --check sample --check-lines 0 --steps 10 --warmup 1"
for i in 1 2 3 4 5 6; do
python bench.py $B --placement-candidates 12 > $O/p$i.json 2> $O/p$i.err
python3 - $O/p$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("process", sys.argv[1][-7:-5], " ".join("%s:%.3f" % (r["device_address"], r["launch_ms"][0]) for r in d["placement"]["candidates"]))
PY
done
