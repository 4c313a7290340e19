#!/bin/bash
# packed fuzz (new test) + three more fresh seeds of it
set -o pipefail
O=gpurun_out/r04ab; mkdir -p $O
for i in 1 2 3; do
timeout -k 10 500 python -m pytest tests/test_gpu_packed.py -m gpu -x -q -k "fuzz" -s > $O/fuzz$i.log 2>&1; rc=$?
grep -i "SEEQ_FUZZ_SEED\|passed\|failed" $O/fuzz$i.log | tr '\n' ' '; echo
[ $rc -ne 0 ] && { tail -30 $O/fuzz$i.log; exit $rc; }
done
exit 0
