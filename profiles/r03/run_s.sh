#!/bin/bash
# Round 3, GPU call s: the whole GPU suite (with the randomized / forced-variant / configs[1] tests).
set -u
O=gpurun_out/r03s; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1150 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
