#!/bin/bash
# round 5, call AC: the seam test on the library from before the fixes (must fail) and now; the fuzz with its multi-pattern pass, eight seeds
out=$PWD/gpurun_out/r05_ac; mkdir -p $out
export TMPDIR=/tmp
cp seeq_amd/lib/libseeq_amd.so /tmp/lib_new.so
cp profiles/r05/ab_libs/libseeq_amd_before_fixes.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "across_a_segment_seam" > $out/pytest_old.log 2>&1; echo "old library: pytest exit $? (expected 1)"; tail -3 $out/pytest_old.log | cut -c1-300
cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "across_a_segment_seam" > $out/pytest_new.log 2>&1; echo "new library: pytest exit $? (expected 0)"; tail -3 $out/pytest_new.log | cut -c1-400
rc=0
for grp in "1 2 3 4" "5 6 7 8"; do
  pids=""
  for i in $grp; do IGNORE_FUZZ_SEGMENTS=$(( i % 2 )) timeout -k 10 900 python3 profiles/ignore_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 40 > $out/fuzz_$i.log 2>&1 & pids="$pids $!"; done
  for p in $pids; do wait $p || rc=1; done
done
for i in 1 2 3 4 5 6 7 8; do echo "--- $i"; head -1 $out/fuzz_$i.log; tail -2 $out/fuzz_$i.log | cut -c1-1500; done
exit $rc
