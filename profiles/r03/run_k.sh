#!/bin/bash
# Round 3, GPU call k: which part of k_pair holds the time (timing-only variants; profiles/time_scan.py checks no result).
set -u
O=gpurun_out/r03k; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
python profiles/time_scan.py exp0 | tee -a $O/exp.txt
SEEQ_PAIR_EXP=2 python profiles/time_scan.py exp2_nogather | tee -a $O/exp.txt
SEEQ_PAIR_EXP=3 python profiles/time_scan.py exp3_nobook | tee -a $O/exp.txt
SEEQ_PAIR_EXP=4 python profiles/time_scan.py exp4_nochk | tee -a $O/exp.txt
SEEQ_PAIR_EXP=5 python profiles/time_scan.py exp5_dbuf | tee -a $O/exp.txt
SEEQ_FUSED_KERNEL=stream python profiles/time_scan.py stream | tee -a $O/exp.txt
done
