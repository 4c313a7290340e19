#!/bin/bash
# k_pair against k_stream on other read lengths (same pattern, --best): is the default the faster one?
set -u
O=gpurun_out/r03bq; mkdir -p $O
for L in 36 75 250; do
  n=$(( 6000000000 / (L + 1) ))
  for k in auto stream; do
    if [ $k = stream ]; then export SEEQ_FUSED_KERNEL=stream; else unset SEEQ_FUSED_KERNEL; fi
    timeout -k 10 300 python bench.py --read-len $L --reads $n --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --check-lines 0 > $O/b_${L}_$k.json 2> $O/b_${L}_$k.err || { tail -3 $O/b_${L}_$k.err; continue; }
    python3 -c "
import json; d=json.load(open('$O/b_${L}_$k.json')); print('L=$L $k', round(d['value']/1e9,2), 'G lines/s', round(d['gb_per_s'],1), 'GB/s', round(d['ms_per_step'],3), 'ms', {a: round(b,3) for a,b in d['device_ms_per_step'].items()}, d['roofline']['kernel'][:8])"
  done
done
