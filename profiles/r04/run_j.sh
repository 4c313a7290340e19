#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_packed.py -m gpu -x -q 2>&1 | tail -3 || exit 1
bash profiles/r04/box_probe.sh
python bench.py --no-per-call --no-cli --no-multi --no-cpu-baseline --no-e2e --check sample --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print(json.dumps(d['packed_scan'], indent=0)[:1500])"
