#!/bin/bash
# round 5, call G: k_verify packs repeats behind partition filters only -- parity subset, then cfg5 / best / all
set -o pipefail
out=gpurun_out/r05_g; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_randomized.py -x -q -m gpu -k "batch_scan_vs_oracle or forced_variants or fastq_records or randomized or golden_cli or chunk_and_tile" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for wl in cfg5 best all count; do
timeout -k 10 600 python bench.py --workload $wl --steps 20 --placement-candidates 1 --first-steps 0 --sections none --check sample > $out/bench_$wl.json 2>$out/bench_$wl.err || { tail -5 $out/bench_$wl.err; exit 1; }
python - $wl <<'PY'
import json,sys
d=json.load(open('gpurun_out/r05_g/bench_%s.json' % sys.argv[1]))
print(sys.argv[1], '%.2f G lines/s %.3f ms' % (d['value']/1e9, d['ms_per_step']), d['device_ms_per_step'], 'frac %.3f whole %.3f' % (d['roofline']['frac'], d['roofline']['whole_step_frac']), d['results']['oracle_check']['result'])
PY
done
