#!/bin/bash
# Round 2, GPU call p: SQ_IGNORE on k_stream (skip variant of the table), two-real-segment parity tests, whole suite.
set -u
O=gpurun_out/r02p; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "ignore_and_convert or every_byte or boundaries or batch_scan or edge_buffers" > $O/pytest_sel.log 2>&1; echo "pytest sel exit $?" >> $O/pytest_sel.log
tail -15 $O/pytest_sel.log
timeout -k 10 800 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -16 $O/pytest_gpu.log
for x in fail convert ignore; do timeout 300 python profiles/fastq_shape_bench.py 5000000 best fastq $x > $O/fastq_$x.json 2> $O/fastq_$x.err; echo "fastq $x exit $?"; cat $O/fastq_$x.json; done
