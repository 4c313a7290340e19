#!/bin/bash
set -u
O=gpurun_out/r03bs; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "forced_variants" > $O/t.log 2>&1; echo "exit $?"; tail -5 $O/t.log
