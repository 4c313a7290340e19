#!/bin/bash
out=$PWD/gpurun_out/r05_aa; mkdir -p $out
export TMPDIR=/tmp
IGNORE_FUZZ_ONLY=31 SEEQ_EXPLAIN=1 timeout -k 10 300 python3 profiles/ignore_fuzz.py 833383578 40 > $out/a.log 2>&1; echo "a $?"; grep -v amdgpu.ids $out/a.log | cut -c1-700 | head -40
