#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04n
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sweep or long_lines or window" 2>&1 | tail -3 || exit 1
python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "stress or long_lines" 2>&1 | tail -3 || exit 1
python profiles/chrom_sweep.py --no-ref > gpurun_out/r04n/sweep_noref.jsonl 2> gpurun_out/r04n/sweep_noref.txt; tail -30 gpurun_out/r04n/sweep_noref.txt
