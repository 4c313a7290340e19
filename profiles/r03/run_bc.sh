#!/bin/bash
set -u
O=gpurun_out/r03bc; mkdir -p $O
t0=$(date +%s)
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "exit $? in $(( $(date +%s) - t0 )) s"; tail -3 $O/bench_default.err
python3 -c "
import json; d=json.load(open('$O/bench_default.json'))
print(round(d['value']/1e9,2),'G lines/s', round(d['ms_per_step'],3),'ms', d['roofline']['avg_launch_ms'], d['roofline']['traffic'])
print(json.dumps(d.get('multi_pattern'), indent=1))"
