#!/bin/bash
out=$PWD/gpurun_out/r05_al; mkdir -p $out
export TMPDIR=/tmp
cp seeq_amd/lib/libseeq_amd.so /tmp/lib_new.so
cp profiles/r05/ab_libs/libseeq_amd_debug.so seeq_amd/lib/libseeq_amd.so
SEEQ_EXPLAIN=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "fasta_header_hit_behind" > $out/pytest_old.log 2>&1; echo "library before the fix: pytest exit $? (expected 1)"; grep "^seeq plan" $out/pytest_old.log | sort | uniq -c | cut -c1-200; grep -c "^J" $out/pytest_old.log; grep "^J" $out/pytest_old.log | head -5; grep "^E" $out/pytest_old.log | sed -n 101,108p
cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so
