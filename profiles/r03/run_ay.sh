#!/bin/bash
set -u
O=gpurun_out/r03ay; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py -m gpu -x -q -k "multi_pattern or packed" > $O/t.log 2>&1; echo "exit $?"; tail -25 $O/t.log
