#!/bin/bash
# A campaign of fresh-seed fuzz runs of the graded tests (short reads, long lines) on the final build: seeds are printed, a
# failing one can be replayed with SEEQ_FUZZ_SEED.
set -u
O=gpurun_out/r03bm; mkdir -p $O
fail=0
for i in $(seq 1 12); do
  timeout -k 10 600 python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "fresh_seed" > $O/t_$i.log 2>&1; rc=$?
  echo "round $i: exit $rc $(grep -h 'SEEQ_FUZZ_SEED' $O/t_$i.log | tr '\n' ' ')"
  if [ $rc -ne 0 ]; then fail=1; grep -h "AssertionError: (" $O/t_$i.log | head -3; fi
done
echo "campaign fail=$fail"
