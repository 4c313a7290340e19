#!/bin/bash
# round 5, call Z: U / u as bases in k_stream's alphabet check -- the new tests on the library from before the fixes (must fail) and on the current one, the fuzz again
out=$PWD/gpurun_out/r05_z; mkdir -p $out
export TMPDIR=/tmp
cp seeq_amd/lib/libseeq_amd.so /tmp/lib_new.so
cp profiles/r05/ab_libs/libseeq_amd_before_fixes.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "begin_with_their_tile or direct_regions or u_and_lower" > $out/pytest_old.log 2>&1; echo "old library: pytest exit $? (expected 1)"; tail -8 $out/pytest_old.log | cut -c1-300
cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_randomized.py -q -m gpu -k "begin_with_their_tile or direct_regions or u_and_lower or kinds" > $out/pytest_new.log 2>&1; echo "new library: pytest exit $? (expected 0)"; tail -6 $out/pytest_new.log | cut -c1-400
rc=0
for grp in "1 2 3 4" "5 6 7 8"; do
  pids=""
  for i in $grp; do timeout -k 10 900 python3 profiles/ignore_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 40 > $out/fuzz_$i.log 2>&1 & pids="$pids $!"; done
  for p in $pids; do wait $p || rc=1; done
done
for i in 1 2 3 4 5 6 7 8; do echo "--- $i"; head -1 $out/fuzz_$i.log; tail -2 $out/fuzz_$i.log | cut -c1-1500; done
exit $rc
