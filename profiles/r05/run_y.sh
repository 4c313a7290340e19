#!/bin/bash
out=$PWD/gpurun_out/r05_y; mkdir -p $out
export TMPDIR=/tmp
IGNORE_FUZZ_ONLY=6 SEEQ_EXPLAIN=1 timeout -k 10 300 python3 profiles/ignore_fuzz.py 457236513 40 > $out/a.log 2>&1; echo "a $?"; grep -v amdgpu.ids $out/a.log | cut -c1-600 | head -40
IGNORE_FUZZ_ONLY=1 timeout -k 10 300 python3 profiles/ignore_fuzz.py 464815435 40 > $out/b.log 2>&1; echo "b $?"; grep -v amdgpu.ids $out/b.log | cut -c1-600 | head -24
cp seeq_amd/lib/libseeq_amd.so /tmp/lib_new.so
cp profiles/r05/ab_libs/libseeq_amd_before_fixes.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "begin_with_their_tile or direct_regions" > $out/pytest_old.log 2>&1; echo "old library: pytest exit $? (expected 1, two failures)"; tail -4 $out/pytest_old.log | cut -c1-300
cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "begin_with_their_tile or direct_regions" > $out/pytest_new.log 2>&1; echo "new library: pytest exit $? (expected 0)"; tail -3 $out/pytest_new.log
