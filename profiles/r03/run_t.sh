#!/bin/bash
# Round 3, GPU call t: the reference's published sweep on its own input shape -- where does every cell land today?
set -u
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 1100 python profiles/chrom_sweep.py > $O/sweep.jsonl 2> $O/sweep.txt; echo "exit $?" >> $O/sweep.txt
cat $O/sweep.txt
