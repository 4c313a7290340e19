#!/bin/bash
# round 5, call B: k_pair's dirty mode (a wave that met a tile with foreign bytes walks on without the fast check) -- parity, FASTQ shape, headline A/B
set -o pipefail
out=gpurun_out/r05_b; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_randomized.py -x -q -m gpu -k "fastq_records_on_the_pair_walk or every_byte_value_alone or ignore_and_convert_with_foreign or batch_scan_vs_oracle or forced_variants or chunk_and_tile or edge_buffers" > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for nd in fail convert ignore; do
  timeout -k 10 300 python profiles/fastq_shape_bench.py 25000000 best fastq $nd > $out/fastq_${nd}.json 2>$out/fastq_${nd}.err || { tail -5 $out/fastq_${nd}.err; exit 1; }
done
cat $out/fastq_*.json
timeout -k 10 600 python bench.py --steps 30 --placement-candidates 4 --no-cpu-baseline --no-e2e --no-per-call --no-packed --no-cli --no-multi --no-fastq --check sample > $out/bench_best.json 2>$out/bench_best.err || { tail -5 $out/bench_best.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_b/bench_best.json'))
print('headline', d['value']/1e9, 'G lines/s', d['ms_per_step'], 'ms', d['device_ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], [ (c['forward_ms']) for c in d['placement']['candidates']])
PY
