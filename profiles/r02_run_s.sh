#!/bin/bash
# Round 2, GPU call s: k_exact1 at 6 waves/SIMD (<= 80 VGPRs) instead of 4 (126 VGPRs): default workload and cfg5.
set -u
O=gpurun_out/r02s; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0"
for w in best all cfg5; do
  timeout -k 10 300 python bench.py --workload $w $B > $O/bench_$w.json 2> $O/bench_$w.err
  python3 -c "
import json; d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],3), d['device_ms_per_step'], round(d['roofline']['avg_launch_ms'],4))"
done
