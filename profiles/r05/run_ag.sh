#!/bin/bash
out=$PWD/gpurun_out/r05_ag; mkdir -p $out
export TMPDIR=/tmp
SEEQ_EXPLAIN=1 timeout -k 10 300 python3 profiles/r05/repro_fasta_header.py > $out/a.log 2>&1; echo "exit $?"; grep -v "amdgpu.ids\|^seeq plan" $out/a.log | cut -c1-220; grep "^seeq plan" $out/a.log | sort | uniq -c | sort -rn | cut -c1-330 | head -12
