#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04o
python -m pytest tests -m gpu -x -q > gpurun_out/r04o/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04o/pytest.log
[ $rc -ne 0 ] && exit $rc
SEEQ_EXPLAIN=1 python bench.py --no-per-call --no-cli --no-cpu-baseline --no-e2e --check sample --check-lines 0 --steps 1 --warmup 0 --reads 2000000 2> gpurun_out/r04o/explain.txt > /dev/null
sort gpurun_out/r04o/explain.txt | uniq -c | sort -rn | head -12
bash profiles/r04/box_probe.sh
