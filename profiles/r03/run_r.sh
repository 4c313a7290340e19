#!/bin/bash
set -u
O=gpurun_out/r03r; mkdir -p $O
for rep in 1 2 3; do
python profiles/time_scan.py verify_best 100000000 10 best | tee -a $O/ab.txt
SEEQ_NO_VERIFY=1 python profiles/time_scan.py exact1_best 100000000 10 best | tee -a $O/ab.txt
done
