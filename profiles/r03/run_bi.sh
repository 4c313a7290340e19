#!/bin/bash
set -u
O=gpurun_out/r03bi; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_randomized.py -m gpu -x -q -k "long or sweep or many_records or mixed_reads or window_walk" > $O/t.log 2>&1; rc=$?; echo "tests exit $rc"; tail -6 $O/t.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 profiles/chrom_sweep.py --cells 20:3,20:5,27:8,34:10,42:14,42:15 --no-ref > $O/sweep.jsonl 2> $O/sweep.err; echo "sweep exit $?"; tail -9 $O/sweep.err
SEEQ_NO_LEADERS=1 timeout -k 10 600 python3 profiles/chrom_sweep.py --cells 20:5,42:14 --no-ref > $O/sweep_nl.jsonl 2> $O/sweep_nl.err; tail -3 $O/sweep_nl.err
python3 - <<'PY'
import json
a = {(d["m"], d["k"]): d for d in map(json.loads, open("gpurun_out/r03bi/sweep.jsonl"))}
b = {(d["m"], d["k"]): d for d in map(json.loads, open("gpurun_out/r03bi/sweep_nl.jsonl"))}
for key in b:
    print(key, "records", a[key].get("records"), b[key].get("records"), "ms", a[key].get("gpu_ms"), b[key].get("gpu_ms"))
PY
