#!/bin/bash
set -u
O=gpurun_out/r03ao; mkdir -p $O
timeout -k 10 800 python3 profiles/multi_bench.py 10000000 > $O/multi.jsonl 2> $O/multi.err; echo "exit $?"; tail -5 $O/multi.err; cat $O/multi.jsonl
