#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04af
timeout -k 10 1000 python -m pytest tests/test_shard_gloo.py tests/test_gpu_randomized.py -m gpu -x -q -k "shard or bench or segments or rccl or dry or PAIR_PF" > gpurun_out/r04af/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r04af/pytest.log
[ $rc -ne 0 ] && exit $rc
B="--no-cpu-baseline --no-e2e --no-per-call --no-cli --no-packed --no-multi --no-fastq --check sample --check-lines 0 --placement-candidates 2 --steps 20"
SEEQ_PAIR_PF=1 python bench.py $B | python3 -c "
import json,sys
d=json.load(sys.stdin); print('PF=1', d['ms_per_step'], [c['launch_ms'] for c in d['placement']['candidates']])"
python bench.py $B | python3 -c "
import json,sys
d=json.load(sys.stdin); print('PF=0', d['ms_per_step'], [c['launch_ms'] for c in d['placement']['candidates']])"
