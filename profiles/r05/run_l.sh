#!/bin/bash
# round 5, call L: SQ_IGNORE on k_pair (line markers) -- the parity tests that exercise skipped bytes, then the FASTQ shape under the three modes
set -o pipefail
out=$PWD/gpurun_out/r05_l; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_randomized.py -x -q -m gpu -k "fastq_records or every_byte_value or ignore_and_convert or batch_scan_vs_oracle or edge_buffers or fuzz_fresh_seed or forced_variants or chunk_and_tile or python_module or filematch or multi_pattern" > $out/pytest.log 2>&1 || { tail -50 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for nd in ignore fail convert; do
  timeout -k 10 300 python profiles/fastq_shape_bench.py 25000000 best fastq $nd > $out/fastq_${nd}.json 2>$out/fastq_${nd}.err || { tail -5 $out/fastq_${nd}.err; exit 1; }
done
cat $out/fastq_*.json
