#!/bin/bash
set -u
O=gpurun_out/r03bd; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_pattern" > $O/t.log 2>&1; echo "exit $?"; tail -25 $O/t.log
