#!/bin/bash
# round 5, call AB: prev_hit_line reflects coverage under SQ_IGNORE -- the failing buffer, the parity tests around segments, eight fresh fuzz seeds with 64 KiB segments on EVERY buffer
out=$PWD/gpurun_out/r05_ab; mkdir -p $out
export TMPDIR=/tmp
IGNORE_FUZZ_ONLY=31 timeout -k 10 300 python3 profiles/ignore_fuzz.py 833383578 40 > $out/a.log 2>&1; echo "replay $?"; tail -1 $out/a.log | cut -c1-300
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_randomized.py tests/test_shard_gloo.py -x -q -m gpu -k "fastq or segment or seam or kinds or foreign or fuzz_fresh or begin_with" > $out/pytest.log 2>&1; echo "pytest $?"; tail -3 $out/pytest.log
rc=0
for grp in "1 2 3 4" "5 6 7 8"; do
  pids=""
  for i in $grp; do IGNORE_FUZZ_SEGMENTS=$(( i % 2 )) timeout -k 10 900 python3 profiles/ignore_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 40 > $out/fuzz_$i.log 2>&1 & pids="$pids $!"; done
  for p in $pids; do wait $p || rc=1; done
done
for i in 1 2 3 4 5 6 7 8; do echo "--- $i"; head -1 $out/fuzz_$i.log; tail -2 $out/fuzz_$i.log | cut -c1-1500; done
exit $rc
