#!/bin/bash
# round 5, call E: k_verify packs the repeats of its range away (full waves) -- the whole GPU suite, then cfg5 and the headline
set -o pipefail
out=gpurun_out/r05_e; mkdir -p $out
s=$(date +%s)
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log; echo "suite: $(( $(date +%s) - s )) s"
for wl in cfg5 best all; do
timeout -k 10 600 python bench.py --workload $wl --steps 20 --placement-candidates 4 --first-steps 0 --sections none --check sample > $out/bench_$wl.json 2>$out/bench_$wl.err || { tail -5 $out/bench_$wl.err; exit 1; }
python - $wl <<'PY'
import json,sys
d=json.load(open('gpurun_out/r05_e/bench_%s.json' % sys.argv[1]))
print(sys.argv[1], '%.2f G lines/s %.3f ms' % (d['value']/1e9, d['ms_per_step']), d['device_ms_per_step'], 'frac %.3f whole %.3f' % (d['roofline']['frac'], d['roofline']['whole_step_frac']), d['placement']['probe_forward_ms'], d['results']['oracle_check']['result'])
PY
done
