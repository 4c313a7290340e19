#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04f
Q="--no-per-call --no-packed --no-cli --no-multi --steps 10 --warmup 3"
python -m pytest tests/test_gpu_randomized.py tests/test_gpu_packed.py -m gpu -x -q 2>&1 | tail -2 || exit 1
for wl in best all; do
  SEEQ_ORDER=old bash profiles/quick_bench.sh "order_old_$wl" $Q --workload $wl | tee -a gpurun_out/r04f/ab.txt
  bash profiles/quick_bench.sh "order_new_$wl" $Q --workload $wl | tee -a gpurun_out/r04f/ab.txt
  SEEQ_VERIFY_OCC=lo bash profiles/quick_bench.sh "order_new_occlo_$wl" $Q --workload $wl | tee -a gpurun_out/r04f/ab.txt
done
bash profiles/r04/run_c.sh 2>&1 | grep -A12 "== new best"
