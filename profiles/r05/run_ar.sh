#!/bin/bash
out=$PWD/gpurun_out/r05_ar; mkdir -p $out
export TMPDIR=/tmp
SEEQ_EXPLAIN=1 timeout -k 10 300 python3 profiles/chrom_sweep.py --no-ref --cells 34:6,34:7,34:8,34:9,42:9,42:12,27:6,27:5,20:4 > $out/a.jsonl 2> $out/a.txt
grep "^seeq plan" $out/a.txt | sort -u | cut -c1-420
