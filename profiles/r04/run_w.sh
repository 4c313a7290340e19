#!/bin/bash
# packed: 32 Mi vs 16 Mi reads per segment (quad walk), then the packed tests
set -o pipefail
O=gpurun_out/r04w; mkdir -p $O
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-multi --no-fastq --check sample --check-lines 0"
for v in 33554432 16777216 67108864; do
  SEEQ_PACKED_SEG_READS=$v timeout -k 10 200 python bench.py $B > $O/seg$v.json 2> $O/seg$v.err || exit 1
  python3 -c "
import json
d=json.load(open('$O/seg$v.json'))
p=d['packed_scan']; print('seg=$v', round(p['lines_per_s']/1e9,2), 'G lines/s', round(p['ms_per_step'],3), 'ms  walk', round(p['scan_kernel_ms_per_launch'],4), 'x', p['scan_kernel_launches_per_step'], p['identical_to_ascii_run'], p['walk_table'][:4])"
done
timeout -k 10 600 python -m pytest tests/test_gpu_packed.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
exit $rc
