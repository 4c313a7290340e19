#!/bin/bash
# packed walk over the quad table (four bases per gather): parity tests, then A/B against the pair table in the bench's packed section
set -o pipefail
O=gpurun_out/r04u; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_packed.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-multi --no-fastq --check sample --check-lines 0"
for v in 1 0; do
  SEEQ_PACKED_QUAD=$v timeout -k 10 200 python bench.py $B > $O/packed_quad$v.json 2> $O/packed_quad$v.err || exit 1
  python3 -c "
import json
d=json.load(open('$O/packed_quad$v.json'))
print('quad=$v', json.dumps(d['packed_scan']))"
done
