#!/bin/bash
# Whole GPU suite + smoke on the committed build.
set -o pipefail
mkdir -p gpurun_out/r04r
python -m pytest tests -m gpu -x -q > gpurun_out/r04r/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04r/pytest.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
