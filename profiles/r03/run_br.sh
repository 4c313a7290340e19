#!/bin/bash
set -u
O=gpurun_out/r03br; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_randomized.py -m gpu -x -q -k "long or sweep or many_records or mixed_reads or window_walk" > $O/t.log 2>&1; rc=$?; echo "tests exit $rc"; tail -5 $O/t.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 profiles/r03/best_dense.py 2>&1 | tail -8
