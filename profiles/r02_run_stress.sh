#!/bin/bash
# Round 2: one scan context reused over 60 random buffers x 5 scans, small segments, tiny workspaces -- against the oracle.
set -u
O=gpurun_out/r02stress; mkdir -p $O
for cfg in "65536 1" "65536 2" "262144 3" "4026531840 4" "65536 5"; do set -- $cfg
  SEEQ_SEGMENT_BYTES=$1 timeout -k 10 200 python profiles/stress_reuse.py $2 > $O/stress_$1_$2.txt 2>&1; echo "seg $1 seed $2: exit $? $(grep -a -E 'stress OK|Error|error|fault' $O/stress_$1_$2.txt | tail -2)"
  if grep -a -q "Memory access fault" $O/stress_$1_$2.txt; then echo "GPU FAULT -- stopping"; exit 1; fi
done
