#!/bin/bash
# round 5, call S: replay of the three buffers the SQ_IGNORE fuzz failed on (call R), with a report on the lines that differ
out=$PWD/gpurun_out/r05_s; mkdir -p $out
export TMPDIR=/tmp
IGNORE_FUZZ_ONLY=27 SEEQ_EXPLAIN=1 timeout -k 10 300 python3 profiles/ignore_fuzz.py 395613376 40 > $out/a.log 2>&1
IGNORE_FUZZ_ONLY=27 timeout -k 10 300 python3 profiles/ignore_fuzz.py 395621381 40 > $out/b.log 2>&1
IGNORE_FUZZ_ONLY=30 timeout -k 10 300 python3 profiles/ignore_fuzz.py 409390223 40 > $out/c.log 2>&1
for f in a b c; do echo "=== $f"; grep -v "amdgpu.ids" $out/$f.log | cut -c1-700 | head -60; done
