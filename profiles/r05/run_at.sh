#!/bin/bash
# round 5, call AT: would k_pair's long-line variant win on cells whose pair automaton flags 2.4 - 2.6 positions per KB?  (a library with the limit at 3 per KB)
out=$PWD/gpurun_out/r05_at; mkdir -p $out
export TMPDIR=/tmp
cp seeq_amd/lib/libseeq_amd.so /tmp/lib_new.so
cp profiles/r05/ab_libs/libseeq_amd_thr.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 300 python3 profiles/chrom_sweep.py --no-ref --cells 20:4,20:5,42:8,42:9 2>&1 | grep -v amdgpu
cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 300 python3 profiles/chrom_sweep.py --no-ref --cells 20:4,20:5,42:8,42:9 2>&1 | grep -v amdgpu
