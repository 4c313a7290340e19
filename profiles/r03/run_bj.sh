#!/bin/bash
set -u
O=gpurun_out/r03bj; mkdir -p $O
timeout -k 10 1150 python3 profiles/chrom_sweep.py > $O/sweep.jsonl 2> $O/sweep.err; echo "sweep exit $?"; tail -26 $O/sweep.err
