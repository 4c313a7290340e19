#!/bin/bash
# round 5, call R: the SQ_IGNORE / dirty-mode fuzz (profiles/ignore_fuzz.py), four fresh seeds, two processes at a time
out=$PWD/gpurun_out/r05_r; mkdir -p $out
export TMPDIR=/tmp
rc=0
for pair in "1 2" "3 4"; do
  pids=""
  for i in $pair; do timeout -k 10 900 python3 profiles/ignore_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 40 > $out/fuzz_$i.log 2>&1 & pids="$pids $!"; done
  for p in $pids; do wait $p || rc=1; done
done
for i in 1 2 3 4; do echo "--- $i"; head -1 $out/fuzz_$i.log; tail -3 $out/fuzz_$i.log | cut -c1-1500; done
exit $rc
