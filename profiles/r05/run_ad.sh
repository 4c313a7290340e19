#!/bin/bash
# round 5, call AD: the CLI against the reference binary on files of line kinds, random options -- three seeds one after the other (four CLI pairs at a time each)
out=$PWD/gpurun_out/r05_ad; mkdir -p $out
export TMPDIR=/tmp
rc=0
for i in 1 2 3; do
  timeout -k 10 600 python3 profiles/cli_diff_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 6 24 > $out/cli_$i.log 2>&1 || rc=1
  head -1 $out/cli_$i.log; tail -12 $out/cli_$i.log | cut -c1-900
done
exit $rc
