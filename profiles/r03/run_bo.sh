#!/bin/bash
set -u
O=gpurun_out/r03bo; mkdir -p $O
for x in fail convert ignore; do timeout 300 python profiles/fastq_shape_bench.py 5000000 best fastq $x > $O/fastq_$x.json 2> $O/fastq_$x.err; python3 -c "
import json; d=json.load(open('$O/fastq_$x.json')); print('$x', {k: d[k] for k in d if k in ('lines_per_s','gb_per_s','kernel','matching_lines')})"; done
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests exit $?"; tail -4 $O/gpu_tests.log
