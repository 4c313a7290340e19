#!/bin/bash
# round 5, call AM: campaigns on the build with the six fixes -- the kinds-of-lines fuzz (16 fresh seeds, four processes at a time), round 2's extended fuzz (read-length, long lines,
# foreign bytes: three seed sets each), the packed fuzz, the CLI differential fuzz
out=$PWD/gpurun_out/r05_am; mkdir -p $out
export TMPDIR=/tmp
rc=0
for grp in "1 2 3 4" "5 6 7 8" "9 10 11 12" "13 14 15 16"; do
  pids=""
  for i in $grp; do IGNORE_FUZZ_SEGMENTS=$(( i % 2 )) timeout -k 10 900 python3 profiles/ignore_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 40 > $out/fuzz_$i.log 2>&1 & pids="$pids $!"; done
  for p in $pids; do wait $p || rc=1; done
  echo "group $grp done rc=$rc"
done
for i in $(seq 1 16); do head -1 $out/fuzz_$i.log; tail -1 $out/fuzz_$i.log | cut -c1-400; done
pids=""
FUZZ_SEED=700 timeout -k 10 900 python3 profiles/extended_fuzz.py > $out/ext_a.log 2>&1 & pids="$pids $!"
FUZZ_SEED=712 FUZZ_FOREIGN=0.3 timeout -k 10 900 python3 profiles/extended_fuzz.py > $out/ext_b.log 2>&1 & pids="$pids $!"
FUZZ_SEED=724 timeout -k 10 900 python3 profiles/extended_fuzz.py long > $out/ext_c.log 2>&1 & pids="$pids $!"
FUZZ_SEED=736 FUZZ_FOREIGN=0.3 timeout -k 10 900 python3 profiles/extended_fuzz.py long > $out/ext_d.log 2>&1 & pids="$pids $!"
for p in $pids; do wait $p || rc=1; done
for f in a b c d; do tail -1 $out/ext_$f.log | cut -c1-300; done
for i in 1 2 3; do timeout -k 10 300 python -m pytest tests/test_gpu_packed.py -q -m gpu -k packed_fuzz -s 2>&1 | grep -a "SEEQ_FUZZ_SEED\|passed\|failed" | tr '\n' ' '; echo; done
timeout -k 10 600 python3 profiles/cli_diff_fuzz.py $(( $(date +%s) % 1000000007 )) 8 24 > $out/cli.log 2>&1 || rc=1; head -1 $out/cli.log; tail -1 $out/cli.log | cut -c1-400
exit $rc
