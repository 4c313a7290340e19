#!/bin/bash
set -u
O=gpurun_out/r03bp; mkdir -p $O
for k in auto stream; do
  [ $k = stream ] && export SEEQ_FUSED_KERNEL=stream
  timeout 300 python profiles/fastq_shape_bench.py 5000000 best fasta fail > $O/fasta_$k.json 2> $O/fasta_$k.err; python3 -c "
import json; d=json.load(open('$O/fasta_$k.json')); print('$k', {k: d[k] for k in d if k in ('lines_per_s','gb_per_s','kernel','matching_lines','lines')})"
done
