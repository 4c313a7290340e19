#!/bin/bash
# round 5, call J: binary search in k_stream_bounds, the long-line filter thresholds for the new Myers speed, parallel CLI tests -- the whole suite, the whole sweep
set -o pipefail
out=$PWD/gpurun_out/r05_j; mkdir -p $out
s=$(date +%s)
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=12 > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -18 $out/pytest.log; echo "suite: $(( $(date +%s) - s )) s"
timeout -k 10 900 python profiles/chrom_sweep.py --no-ref > $out/sweep_noref.jsonl 2>$out/sweep_noref.txt || { tail -5 $out/sweep_noref.txt; exit 1; }
cat $out/sweep_noref.txt
