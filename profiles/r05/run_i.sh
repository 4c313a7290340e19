#!/bin/bash
# round 5, call I: k_stream's Myers mode with the lean steps on words without a flagged byte -- parity (sweep cells, long-line fuzz), then the cells' times
set -o pipefail
out=$PWD/gpurun_out/r05_i; mkdir -p $out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu -k "published_sweep or long_lines or stream_fuzz_long or myers or chromosome or NO_MYERS" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 600 python profiles/chrom_sweep.py --cells 20:4,20:5,27:6,27:8,34:9,34:10,42:9,42:12,42:15 --no-ref > $out/sweep_noref.jsonl 2>$out/sweep_noref.txt || { tail -5 $out/sweep_noref.txt; exit 1; }
cat $out/sweep_noref.txt
timeout -k 10 600 python profiles/chrom_sweep.py --cells 20:4,27:7,42:9 > $out/sweep_ref.jsonl 2>$out/sweep_ref.txt || { tail -5 $out/sweep_ref.txt; exit 1; }
cat $out/sweep_ref.txt
