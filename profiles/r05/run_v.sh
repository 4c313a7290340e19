#!/bin/bash
out=$PWD/gpurun_out/r05_v; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python3 profiles/r05/replay_buf.py profiles/r05/fuzzbuf_790497392_19.bin AACAAAAAAAAA 1 > $out/a.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $out/a.log | cut -c1-250
