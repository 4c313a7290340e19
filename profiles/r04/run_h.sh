#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04h
python -m pytest tests/test_shard_gloo.py -m gpu -x -q 2>&1 | tail -3 || exit 1
python bench.py > gpurun_out/r04h/bench_default.json 2> gpurun_out/r04h/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04h/bench_default.json") if l.startswith("{")][-1])
print(d["value"]/1e9, d["ms_per_step"], d["device_ms_per_step"], d["roofline"]["frac"], d["roofline"]["whole_step_frac"])
print(json.dumps(d["per_step"]))
print(json.dumps(d["results"]["oracle_check"]))
print(json.dumps(d["hbm_per_rank"]))
PY
