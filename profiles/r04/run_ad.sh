#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04ad
timeout -k 10 900 python -m pytest tests/test_gpu_randomized.py -m gpu -x -q -k "PAIR_PF or EMIT_ALL" > gpurun_out/r04ad/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r04ad/pytest.log
exit $rc
