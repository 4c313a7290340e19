#!/bin/bash
# round 5, call T: after the fix of the known-start rule -- the three failing buffers replayed, then six fresh seeds of the SQ_IGNORE fuzz, three processes at a time
out=$PWD/gpurun_out/r05_t; mkdir -p $out
export TMPDIR=/tmp
IGNORE_FUZZ_ONLY=27 timeout -k 10 300 python3 profiles/ignore_fuzz.py 395613376 40 > $out/a.log 2>&1; echo "a $?"; tail -1 $out/a.log | cut -c1-300
IGNORE_FUZZ_ONLY=27 timeout -k 10 300 python3 profiles/ignore_fuzz.py 395621381 40 > $out/b.log 2>&1; echo "b $?"; tail -1 $out/b.log | cut -c1-300
IGNORE_FUZZ_ONLY=30 timeout -k 10 300 python3 profiles/ignore_fuzz.py 409390223 40 > $out/c.log 2>&1; echo "c $?"; tail -1 $out/c.log | cut -c1-300
rc=0
for grp in "1 2 3" "4 5 6"; do
  pids=""
  for i in $grp; do timeout -k 10 900 python3 profiles/ignore_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 40 > $out/fuzz_$i.log 2>&1 & pids="$pids $!"; done
  for p in $pids; do wait $p || rc=1; done
done
for i in 1 2 3 4 5 6; do echo "--- $i"; head -1 $out/fuzz_$i.log; tail -2 $out/fuzz_$i.log | cut -c1-1200; done
exit $rc
