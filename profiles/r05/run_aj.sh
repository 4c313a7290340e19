#!/bin/bash
out=$PWD/gpurun_out/r05_aj; mkdir -p $out
export TMPDIR=/tmp
cp seeq_amd/lib/libseeq_amd.so /tmp/lib_new.so
cp profiles/r05/ab_libs/libseeq_amd_debug.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 500 python3 profiles/r05/bisect_buf.py profiles/r05/fuzzbuf_fasta.bin CAACCCCAACACCACAACCAAAAA 4 6 1 > $out/a.log 2>&1; echo "exit $?"
cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so
grep -v amdgpu.ids $out/a.log | grep -n "=== debug" | tail -1
awk '/=== debug/{f=1} f' $out/a.log | cut -c1-200 | head -60
