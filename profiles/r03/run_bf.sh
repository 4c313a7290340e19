#!/bin/bash
set -u
O=gpurun_out/r03bf; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "published_sweep" > $O/t.log 2>&1; echo "exit $?"; tail -25 $O/t.log
