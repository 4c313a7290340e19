#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04am
timeout -k 10 600 python -m pytest tests/test_gpu_packed.py -m gpu -x -q -k "text_alloc" > gpurun_out/r04am/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r04am/pytest.log
[ $rc -ne 0 ] && exit $rc
python3 - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
from seeq_amd import device as dev
t=time.time(); b = dev.TextBuffer(15_100_000_000, candidates=8); print("TextBuffer(15.1 GB, 8 candidates): %.2f s, probe ms" % (time.time()-t), [round(x,3) for x in b.probe_ms], "chosen", min(range(len(b.probe_ms)), key=lambda i: b.probe_ms[i]))
b.free()
PY
