#!/bin/bash
# fresh-seed campaigns on the final build: the read-length fuzz, the long-line fuzz, the long-line stress (3 variants x 70), the extended fuzz
mkdir -p gpurun_out/r04fuzz
for i in 1 2 3 4; do
  python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "fresh_seed" -s 2>&1 | grep -i "SEEQ_FUZZ_SEED\|passed\|failed\|Error\|assert" | tr '\n' ' ' | tee -a gpurun_out/r04fuzz/campaign.txt; echo | tee -a gpurun_out/r04fuzz/campaign.txt
done
timeout -k 10 500 python profiles/extended_fuzz.py 2>&1 | tail -6 | tee -a gpurun_out/r04fuzz/campaign.txt
