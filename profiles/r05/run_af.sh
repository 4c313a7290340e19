#!/bin/bash
out=$PWD/gpurun_out/r05_af; mkdir -p $out
export TMPDIR=/tmp
IGNORE_FUZZ_ONLY=38 SEEQ_EXPLAIN=1 timeout -k 10 300 python3 profiles/ignore_fuzz.py 836473034 40 > $out/a.log 2>&1; echo "a $?"; grep -v amdgpu.ids $out/a.log | grep -v "^seeq plan" | cut -c1-900 | head -14; grep "^seeq plan" $out/a.log | sort | uniq -c | cut -c1-260 | head -8
IGNORE_FUZZ_ONLY=37 timeout -k 10 300 python3 profiles/ignore_fuzz.py 869102446 40 > $out/b.log 2>&1; echo "b $?"; grep -v amdgpu.ids $out/b.log | cut -c1-900 | head -12
