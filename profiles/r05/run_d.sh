#!/bin/bash
# round 5, call D: the reworked bench.py -- its tests, then the default line as the driver runs it
set -o pipefail
out=gpurun_out/r05_d; mkdir -p $out
timeout -k 10 1000 python -m pytest tests/test_shard_gloo.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
s=$(date +%s)
timeout -k 10 900 python bench.py > $out/bench_default.json 2>$out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
echo "default bench.py: $(( $(date +%s) - s )) s, $(wc -c < $out/bench_default.json) bytes"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_d/bench_default.json'))
print('headline %.2f G lines/s %.3f ms' % (d['value']/1e9, d['ms_per_step']), d['device_ms_per_step'], 'frac %.3f whole %.3f' % (d['roofline']['frac'], d['roofline']['whole_step_frac']))
print('first_allocation', d['first_allocation'])
print('placement', d['placement'])
print('fastq', {m: (round(v['gb_per_s']), v['kernel'], v['ms_per_step']) for m, v in d['fastq_shape']['modes'].items()} if 'modes' in d.get('fastq_shape', {}) else d.get('fastq_shape'))
c=d.get('cfg5'); print('cfg5', {k: c[k] for k in ('value','ms_per_step','device_ms_per_step','whole_step_frac')} if c and 'value' in c else c)
print('check', d['results']['oracle_check'])
print('seconds', d['seconds'], 'tail keys', list(d)[-8:])
PY
