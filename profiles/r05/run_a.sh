#!/bin/bash
# round 5, call A: FASTQ-shaped text on k_pair (dirty tiles from registers, prefix check in k_verify) -- parity first, then the A/B
set -o pipefail
out=gpurun_out/r05_a; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fastq_records_on_the_pair_walk or every_byte_value_alone or ignore_and_convert_with_foreign or batch_scan_vs_oracle" > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for nd in fail convert ignore; do
  timeout -k 10 300 python profiles/fastq_shape_bench.py 25000000 best fastq $nd > $out/fastq_${nd}_new.json 2>$out/fastq_${nd}_new.err || { tail -5 $out/fastq_${nd}_new.err; exit 1; }
  SEEQ_FUSED_KERNEL=stream timeout -k 10 300 python profiles/fastq_shape_bench.py 25000000 best fastq $nd > $out/fastq_${nd}_stream.json 2>$out/fastq_${nd}_stream.err || { tail -5 $out/fastq_${nd}_stream.err; exit 1; }
done
cat $out/fastq_*.json
