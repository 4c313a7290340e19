#!/bin/bash
out=$PWD/gpurun_out/r05_ah; mkdir -p $out
export TMPDIR=/tmp
for v in default SEEQ_NO_LEADERS SEEQ_NO_MYERS SEEQ_NO_WINDOW; do
  if [ $v = default ]; then IGNORE_FUZZ_ONLY=37 timeout -k 10 300 python3 profiles/ignore_fuzz.py 869102446 40 > $out/$v.log 2>&1
  else env $v=1 IGNORE_FUZZ_ONLY=37 timeout -k 10 300 python3 profiles/ignore_fuzz.py 869102446 40 > $out/$v.log 2>&1; fi
  echo "== $v exit $?"; grep "^DIFF\|^EXTRA\|^MISSING\|OK\|FAILED" $out/$v.log | cut -c1-200 | head -8
done
