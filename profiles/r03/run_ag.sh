#!/bin/bash
set -u
O=gpurun_out/r03ag; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_packed.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -25 $O/pytest.log
