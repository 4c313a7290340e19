#!/bin/bash
# Round 3, GPU call u: k_stream's Myers mode -- long-line tests, then the sweep again.
set -u
O=gpurun_out/r03v; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "long or edge_buffers or chunk_and_tile or forced_variants or fuzz_long or batch_scan" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout -k 10 600 python profiles/chrom_sweep.py > $O/sweep.jsonl 2> $O/sweep.txt; echo "exit $?" >> $O/sweep.txt
cat $O/sweep.txt
