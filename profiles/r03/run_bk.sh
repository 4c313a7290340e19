#!/bin/bash
set -u
O=gpurun_out/r03bk; mkdir -p $O
SEEQ_FUZZ_SEED=119900423 timeout -k 10 600 python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "fuzz_long_lines_fresh" > $O/t1.log 2>&1; echo "leaders on: exit $?"; grep -n "AssertionError: (" $O/t1.log | head -3
SEEQ_NO_LEADERS=1 SEEQ_FUZZ_SEED=119900423 timeout -k 10 600 python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "fuzz_long_lines_fresh" > $O/t2.log 2>&1; echo "leaders off: exit $?"; grep -n "AssertionError: (" $O/t2.log | head -3
