#!/bin/bash
# round 5, call F: k_verify's range -- 256 / 512 / 1024 entries per workgroup -- on configs[4] (three-part filter: > half of the entries are repeats) and the headline (a quarter)
set -o pipefail
out=gpurun_out/r05_f; mkdir -p $out
for wl in cfg5 best all; do for vr in 256 512 1024; do
SEEQ_VERIFY_RANGE=$vr timeout -k 10 600 python bench.py --workload $wl --steps 20 --placement-candidates 1 --first-steps 0 --sections none --check-lines 0 > $out/bench_${wl}_$vr.json 2>$out/bench_${wl}_$vr.err || { tail -5 $out/bench_${wl}_$vr.err; exit 1; }
python - $wl $vr <<'PY'
import json,sys
d=json.load(open('gpurun_out/r05_f/bench_%s_%s.json' % (sys.argv[1], sys.argv[2])))
print(sys.argv[1], sys.argv[2], '%.2f G lines/s %.3f ms' % (d['value']/1e9, d['ms_per_step']), d['device_ms_per_step'])
PY
done; done
