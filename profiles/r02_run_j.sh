#!/bin/bash
# Round 2, GPU call j: pipelined ingest (reader thread, lanes, SEEQ_DEVICES) -- CLI / file tests, then the whole GPU suite.
set -u
O=gpurun_out/r02j; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "cli or filematch or python" > $O/pytest_cli.log 2>&1; echo "pytest cli exit $?" >> $O/pytest_cli.log
tail -30 $O/pytest_cli.log
timeout -k 10 500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_cli_golden_outputs > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
