#!/bin/bash
O=gpurun_out/r04ae; mkdir -p $O
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq --check sample --check-lines 0"
for n in 8 4 8 1; do
python bench.py $B --placement-candidates $n > $O/p$n.json 2> $O/p$n.err
python3 - $O/p$n.json $n <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("candidates", sys.argv[2], "value", round(d["value"]/1e9,2), "ms", round(d["ms_per_step"],3), d["per_step"]["ms"], "chosen", d["placement"]["chosen"] if d["placement"] else None)
print("   steps:", d["per_step"]["ms_all"])
PY
done
