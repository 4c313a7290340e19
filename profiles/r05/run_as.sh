#!/bin/bash
# round 5, call AS: SQ_IGNORE on k_pair at 2 M lines with every quality line marked (share 0.5) and with few of them marked (0.1), and 64 KiB / 1 MiB segments
out=$PWD/gpurun_out/r05_as; mkdir -p $out
export TMPDIR=/tmp
rc=0
timeout -k 10 600 python3 profiles/r05/ignore_big.py 500000 0.5 > $out/a.log 2>&1 || rc=1; grep -v amdgpu $out/a.log | cut -c1-250
timeout -k 10 600 python3 profiles/r05/ignore_big.py 500000 0.1 7 > $out/b.log 2>&1 || rc=1; grep -v amdgpu $out/b.log | cut -c1-250
SEEQ_SEGMENT_BYTES=1048576 timeout -k 10 600 python3 profiles/r05/ignore_big.py 200000 0.5 9 > $out/c.log 2>&1 || rc=1; grep -v amdgpu $out/c.log | cut -c1-250
exit $rc
