#!/bin/bash
mkdir -p gpurun_out/r04m
SECONDS=0; python bench.py > gpurun_out/r04m/bench_default.json 2> gpurun_out/r04m/bench_default.err; echo "bench rc=$?"
echo "bench wall seconds: $SECONDS"; tail -3 gpurun_out/r04m/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04m/bench_default.json") if l.startswith("{")][-1])
print(d["value"]/1e9, d["ms_per_step"], d["device_ms_per_step"], d["roofline"]["frac"], d["roofline"]["whole_step_frac"])
print(json.dumps(d.get("fastq_shape"))[:1500])
print(json.dumps(d.get("packed_scan"))[:600])
print({k: (v if not isinstance(v, dict) else "...") for k, v in d.items() if k not in ("config",)}.keys())
PY
bash profiles/r04/box_probe.sh
