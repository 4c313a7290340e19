#!/bin/bash
set -u
O=gpurun_out/r03bn; mkdir -p $O
for x in fail convert ignore; do timeout 300 python profiles/fastq_shape_bench.py 5000000 best fastq $x > $O/fastq_$x.json 2> $O/fastq_$x.err; python3 -c "
import json; d=json.load(open('$O/fastq_$x.json')); print('$x', {k: d[k] for k in d if k in ('lines_per_s','gb_per_s','kernel','ms','ms_per_scan','matching_lines')})"; done
for x in fail convert; do SEEQ_FUSED_KERNEL=stream timeout 300 python profiles/fastq_shape_bench.py 5000000 best fastq $x > $O/fastq_${x}_stream.json 2> $O/fastq_${x}_stream.err; python3 -c "
import json; d=json.load(open('$O/fastq_${x}_stream.json')); print('$x stream', {k: d[k] for k in d if k in ('lines_per_s','gb_per_s','kernel','ms','ms_per_scan','matching_lines')})"; done
