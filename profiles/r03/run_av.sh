#!/bin/bash
set -u
O=gpurun_out/r03av; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "exit $?"; tail -8 $O/gpu_tests.log
