#!/bin/bash
# Round 2, GPU call o: whole GPU suite on the cleaned-up build (k_stream2 / lazy check / overlapped post-pass removed),
# then rocprofv3 kernel stats + PMC passes of the default workload, kernel stats of cfg5.
set -u
O=gpurun_out/r02o; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
TEXT_BYTES_TOTAL=60400000000 timeout -k 10 500 bash profiles/gpu_profile.sh r02_best --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-per-call --check-lines 0 > $O/profile_best.log 2>&1; echo "profile best exit $?"
tail -60 gpurun_out/prof_r02_best/summary.txt
REPO=$PWD; cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_cfg5 -- python3 $REPO/bench.py --workload cfg5 --steps 3 --warmup 1 --no-e2e --no-per-call --no-cpu-baseline --check-lines 0 > $REPO/$O/prof_cfg5.log 2>&1
cd $REPO
find $O gpurun_out/prof_r02_best -name "*.csv" -size +4M -delete
head -16 $O/prof_cfg5/*/*_kernel_stats.csv | cut -c1-220
