#!/bin/bash
# round 4, first GPU call: the graded suite on the new exact pass (k_verify), then old vs new on one box
set -o pipefail
mkdir -p gpurun_out/r04a
python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r04a/pytest.log
[ $rc -ne 0 ] && exit $rc
Q="--no-per-call --no-packed --no-cli --no-multi --steps 10 --warmup 3"
for wl in best count all; do
  SEEQ_VERIFY=old bash profiles/quick_bench.sh "old_$wl" $Q --workload $wl | tee -a gpurun_out/r04a/ab.txt
  bash profiles/quick_bench.sh "new_$wl" $Q --workload $wl | tee -a gpurun_out/r04a/ab.txt
done
SEEQ_VERIFY=old bash profiles/quick_bench.sh "old_cfg5" $Q --workload cfg5 | tee -a gpurun_out/r04a/ab.txt
bash profiles/quick_bench.sh "new_cfg5" $Q --workload cfg5 | tee -a gpurun_out/r04a/ab.txt
