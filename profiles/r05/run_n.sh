#!/bin/bash
# round 5, call N: cheaper SQ_IGNORE bookkeeping in k_pair (non-base mask by one table look-up, line ends with compile-time groups): parity + the FASTQ shape
out=$PWD/gpurun_out/r05_n; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_randomized.py -x -q -m gpu -k "fastq or foreign or fresh_seed or fuzz or ignore or batch_scan" > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $out/pytest.log
for m in fail convert ignore; do
  timeout -k 10 200 python3 profiles/fastq_shape_bench.py 25000000 best fastq $m > $out/fastq_$m.json 2> $out/fastq_$m.err; echo "$m exit $?"; cut -c1-400 $out/fastq_$m.json
done
