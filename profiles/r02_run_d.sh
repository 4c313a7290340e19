#!/bin/bash
# Round 2, GPU call d (after the container was re-created): GPU tests + the default bench line.
set -u
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout -k 10 330 python bench.py --steps 10 --warmup 3 > $O/bench_best.json 2> $O/bench_best.err; echo "bench best exit $?"
head -c 3000 $O/bench_best.json; tail -3 $O/bench_best.err
