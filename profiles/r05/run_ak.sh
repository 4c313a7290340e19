#!/bin/bash
# round 5, call AK: the FASTA jump fix -- the new test on the debug-free library from before (the debug one has the bug: must fail) and now; the three failing buffers; eight fuzz seeds
out=$PWD/gpurun_out/r05_ak; mkdir -p $out
export TMPDIR=/tmp
cp seeq_amd/lib/libseeq_amd.so /tmp/lib_new.so
cp profiles/r05/ab_libs/libseeq_amd_debug.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fasta_header_hit_behind" > $out/pytest_old.log 2>&1; echo "library before the fix: pytest exit $? (expected 1)"; tail -2 $out/pytest_old.log | cut -c1-300
cp /tmp/lib_new.so seeq_amd/lib/libseeq_amd.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fasta" > $out/pytest_new.log 2>&1; echo "new library: pytest exit $? (expected 0)"; tail -2 $out/pytest_new.log | cut -c1-300
IGNORE_FUZZ_ONLY=38 timeout -k 10 300 python3 profiles/ignore_fuzz.py 836473034 40 > $out/a.log 2>&1; echo "a $?"; tail -1 $out/a.log | cut -c1-200
IGNORE_FUZZ_ONLY=4 IGNORE_FUZZ_SEGMENTS=1 timeout -k 10 300 python3 profiles/ignore_fuzz.py 836480934 40 > $out/b.log 2>&1; echo "b $?"; tail -1 $out/b.log | cut -c1-200
IGNORE_FUZZ_ONLY=37 IGNORE_FUZZ_SEGMENTS=1 timeout -k 10 300 python3 profiles/ignore_fuzz.py 869102446 40 > $out/c.log 2>&1; echo "c $?"; tail -1 $out/c.log | cut -c1-200
rc=0
for grp in "1 2 3 4" "5 6 7 8"; do
  pids=""
  for i in $grp; do IGNORE_FUZZ_SEGMENTS=$(( i % 2 )) timeout -k 10 900 python3 profiles/ignore_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 40 > $out/fuzz_$i.log 2>&1 & pids="$pids $!"; done
  for p in $pids; do wait $p || rc=1; done
done
for i in 1 2 3 4 5 6 7 8; do echo "--- $i"; head -1 $out/fuzz_$i.log; tail -2 $out/fuzz_$i.log | cut -c1-1200; done
exit $rc
