#!/bin/bash
# Round 2, first GPU call: GPU tests, the two BASELINE bench lines, configs[4] on the old per-line kernel for comparison,
# and a kernel-trace of configs[4].  Output under gpurun_out/r02a/.
set -u
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_best.json 2> $O/bench_best.err; echo "bench best exit $?"
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-e2e --no-per-call > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "bench cfg5 exit $?"
SEEQ_FUSED_KERNEL=direct timeout 900 python bench.py --workload cfg5 --steps 3 --warmup 1 --no-e2e --no-per-call --no-cpu-baseline --check-lines 200000 > $O/bench_cfg5_kdirect.json 2> $O/bench_cfg5_kdirect.err; echo "bench cfg5 k_direct exit $?"
REPO=$PWD; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_cfg5 -- python3 $REPO/bench.py --workload cfg5 --steps 3 --warmup 1 --no-e2e --no-per-call --no-cpu-baseline --check-lines 0 > $REPO/$O/prof_cfg5.log 2>&1
cd $REPO
find $O -name "*.csv" -size +8M -delete
for f in $O/bench_*.json; do echo "== $f"; head -c 1500 $f; echo; done
tail -3 $O/*.err
