#!/bin/bash
mkdir -p gpurun_out/r04i
Q="--no-per-call --no-packed --no-cli --no-multi --no-cpu-baseline --no-e2e --check sample --check-lines 0"
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 1; /opt/rocm/bin/rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -i "sclk\|power\|perf\|fclk\|mclk" ; echo ---; done ) > gpurun_out/r04i/smi.txt 2>&1 &
python bench.py $Q --steps 1500 --warmup 5 > gpurun_out/r04i/b1.json 2>gpurun_out/r04i/b1.err
wait
python bench.py $Q --steps 50 --warmup 5 --log-clocks > gpurun_out/r04i/b2.json 2>/dev/null
python bench.py $Q --steps 50 --warmup 5 > gpurun_out/r04i/b3.json 2>/dev/null
python - <<'PY'
import json
for f in ("b1","b2","b3"):
    d=json.loads([l for l in open("gpurun_out/r04i/%s.json"%f) if l.startswith("{")][-1])
    print(f, d["ms_per_step"], d["per_step"]["ms"], d["per_step"]["forward_scan_ms"], d["per_step"]["scan_launch_ms_full_segments"], d["per_step"]["gpu_clock_power_during_steps"])
PY
head -60 gpurun_out/r04i/smi.txt
ls /sys/class/drm/card*/device/gpu_metrics 2>/dev/null | head -3
python - <<'PY'
import glob,struct
for f in glob.glob("/sys/class/drm/card*/device/gpu_metrics")[:1]:
    b=open(f,"rb").read()
    print(f, len(b), struct.unpack_from("<HBB", b, 0))
PY
