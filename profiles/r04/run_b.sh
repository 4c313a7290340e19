#!/bin/bash
# old vs new exact pass (SEEQ_VERIFY=old: k_exact1<COUNT> + three-launch scan) on one box
set -o pipefail
mkdir -p gpurun_out/r04b
Q="--no-per-call --no-packed --no-cli --no-multi --steps 10 --warmup 3"
for wl in best count all cfg5; do
  SEEQ_VERIFY=old bash profiles/quick_bench.sh "old_$wl" $Q --workload $wl | tee -a gpurun_out/r04b/ab.txt
  bash profiles/quick_bench.sh "new_$wl" $Q --workload $wl | tee -a gpurun_out/r04b/ab.txt
done
