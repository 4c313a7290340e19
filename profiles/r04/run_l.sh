#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04l
python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "fuzz or forced" 2>&1 | tail -3 || exit 1
for nd in fail convert ignore; do
  python profiles/fastq_shape_bench.py 5000000 best fastq $nd 2>/dev/null | tee -a gpurun_out/r04l/fastq.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['nondna'], round(d['lines_per_s']/1e9,2), 'G lines/s', round(d['ms_per_step'],3), 'ms', d['kernel'], d['matching_lines'], d['oracle_prefix_check'], d['times_ms'])"
done
