#!/bin/bash
# Round 2, GPU call v: newline masks by SWAR + v_dot4 instead of a compare per character -- GPU suite, bench lines.
set -u
O=gpurun_out/r02v; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0"
for w in best count cfg5; do
  timeout -k 10 300 python bench.py --workload $w $B > $O/bench_$w.json 2> $O/bench_$w.err
  python3 -c "
import json; d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],3), d['device_ms_per_step'], round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"
done
for x in fail convert; do timeout 300 python profiles/fastq_shape_bench.py 5000000 best fastq $x > $O/fastq_$x.json 2> $O/fastq_$x.err; python3 -c "
import json; d=json.load(open('$O/fastq_$x.json')); print('fastq $x', round(d['ms_per_step'],3), d['times_ms'])"; done
