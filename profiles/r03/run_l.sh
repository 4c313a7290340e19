#!/bin/bash
# Round 3, GPU call l: the whole GPU suite on the k_pair build.
set -u
O=gpurun_out/r03l; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
