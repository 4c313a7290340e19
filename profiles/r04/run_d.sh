#!/bin/bash
set -o pipefail
python profiles/r04/repro1.py 200000 rec 2>&1 | tail -4 && python profiles/r04/repro1.py 200000 lines 2>&1 | tail -1 && python profiles/r04/repro1.py 200000 match 2>&1 | tail -1 || exit 1
python -m pytest tests/test_gpu_randomized.py tests/test_gpu_packed.py -m gpu -x -q 2>&1 | tail -3 || exit 1
bash profiles/r04/run_b.sh
bash profiles/r04/run_c.sh 2>&1 | grep -A12 "== new"
