#!/bin/bash
# Round 2, GPU call m: multi-pattern scan test, HIP start-up probe, CLI start-up split.
set -u
O=gpurun_out/r02m; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "multi_pattern" > $O/pytest_multi.log 2>&1; echo "pytest exit $?" >> $O/pytest_multi.log
tail -15 $O/pytest_multi.log
profiles/microbench/hip_startup > $O/hip_startup.txt 2>&1; profiles/microbench/hip_startup >> $O/hip_startup.txt 2>&1; cat $O/hip_startup.txt
printf 'ACGT\n' > /dev/shm/seeq_tiny.txt
for i in 1 2 3; do /usr/bin/env time -f "%e s wall" seeq_amd/bin/seeq -z -c ACGT /dev/shm/seeq_tiny.txt 2>&1 | tail -4; done
ls -la seeq_amd/lib/libseeq_amd.so
