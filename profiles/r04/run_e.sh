#!/bin/bash
# k_verify variants on one box: default, fewer waves per SIMD (no spills)
set -o pipefail
mkdir -p gpurun_out/r04e
Q="--no-per-call --no-packed --no-cli --no-multi --steps 10 --warmup 3"
python -m pytest tests/test_gpu_randomized.py -m gpu -x -q 2>&1 | tail -2 || exit 1
for wl in best all cfg5; do
  bash profiles/quick_bench.sh "occ_hi_$wl" $Q --workload $wl | tee -a gpurun_out/r04e/ab.txt
  SEEQ_VERIFY_OCC=lo bash profiles/quick_bench.sh "occ_lo_$wl" $Q --workload $wl | tee -a gpurun_out/r04e/ab.txt
done
