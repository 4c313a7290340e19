#!/bin/bash
set -u
O=gpurun_out/r03bl; mkdir -p $O
timeout -k 10 300 python3 profiles/r03/dbg_lead.py 2>&1 | grep "kernel\|first diffs"
SEEQ_FUZZ_SEED=119900423 timeout -k 10 600 python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "fuzz_long_lines_fresh" > $O/t1.log 2>&1; echo "seed replay: exit $?"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_lines_with_many_hits or sweep or window_walk or many_records" > $O/t2.log 2>&1; echo "leader tests: exit $?"; tail -4 $O/t2.log
for i in 1 2 3; do timeout -k 10 600 python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "fuzz_long_lines_fresh" > $O/t3_$i.log 2>&1; echo "fresh seed $i: exit $?"; grep "SEEQ_FUZZ_SEED\|AssertionError: (" $O/t3_$i.log | head -3; done
