#!/bin/bash
# Round 2, GPU call q: SUB variants without the CSE spills; SQ_IGNORE falls back to the per-line kernel on mostly-foreign text.
set -u
O=gpurun_out/r02q; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "ignore_and_convert or every_byte or boundaries or batch_scan or edge_buffers or fuzz" > $O/pytest_sel.log 2>&1; echo "pytest sel exit $?" >> $O/pytest_sel.log
tail -5 $O/pytest_sel.log
for x in fail convert ignore; do timeout 300 python profiles/fastq_shape_bench.py 5000000 best fastq $x > $O/fastq_$x.json 2> $O/fastq_$x.err; echo "fastq $x exit $?"; cat $O/fastq_$x.json; done
for x in convert ignore; do timeout 300 python profiles/fastq_shape_bench.py 5000000 count fastq $x > $O/fastq_count_$x.json 2> $O/fastq_count_$x.err; cat $O/fastq_count_$x.json; done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0"
timeout -k 10 200 python bench.py $B > $O/bench_best.json 2> $O/bench_best.err; python3 -c "
import json; d=json.load(open('$O/bench_best.json')); print(d['ms_per_step'], d['device_ms_per_step'], d['roofline']['avg_launch_ms'])"
