#!/bin/bash
# round 5, call AN: the CLI differential fuzz with the ingest pipeline cut into small chunks in half of the runs -- three seeds
out=$PWD/gpurun_out/r05_an; mkdir -p $out
export TMPDIR=/tmp
rc=0
for i in 1 2 3; do
  timeout -k 10 900 python3 profiles/cli_diff_fuzz.py $(( ( $(date +%s%N) / 1000 + i * 7919 ) % 1000000007 )) 8 24 > $out/cli_$i.log 2>&1 || rc=1
  head -1 $out/cli_$i.log; tail -8 $out/cli_$i.log | cut -c1-1200
done
exit $rc
