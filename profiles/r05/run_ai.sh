#!/bin/bash
out=$PWD/gpurun_out/r05_ai; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 python3 profiles/r05/bisect_buf.py profiles/r05/fuzzbuf_fasta.bin CAACCCCAACACCACAACCAAAAA 4 6 1 > $out/a.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $out/a.log | cut -c1-260 | tail -24
