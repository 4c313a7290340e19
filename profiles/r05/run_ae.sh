#!/bin/bash
# round 5, call AE: the fuzz with FASTA buffers (a fifth of them), eight seeds
out=$PWD/gpurun_out/r05_ae; mkdir -p $out
export TMPDIR=/tmp
rc=0
for grp in "1 2 3 4" "5 6 7 8"; do
  pids=""
  for i in $grp; do IGNORE_FUZZ_SEGMENTS=$(( i % 2 )) timeout -k 10 900 python3 profiles/ignore_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 40 > $out/fuzz_$i.log 2>&1 & pids="$pids $!"; done
  for p in $pids; do wait $p || rc=1; done
done
for i in 1 2 3 4 5 6 7 8; do echo "--- $i"; head -1 $out/fuzz_$i.log; tail -2 $out/fuzz_$i.log | cut -c1-1500; done
exit $rc
