#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04g
python -m pytest tests -m gpu -x -q > gpurun_out/r04g/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04g/pytest.log
[ $rc -ne 0 ] && exit $rc
Q="--no-per-call --no-packed --no-cli --no-multi --steps 10 --warmup 3"
for wl in best count all cfg5; do
  bash profiles/quick_bench.sh "new_$wl" $Q --workload $wl | tee -a gpurun_out/r04g/ab.txt
done
bash profiles/r04/run_c.sh 2>&1 | grep -A10 "== new best"
